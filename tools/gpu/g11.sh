set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q -k "bundle_adjustment" 2>&1 | tail -15
