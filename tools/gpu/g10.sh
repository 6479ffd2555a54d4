set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/gpu/probe_ba.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_slam_driver_gpu.py -x -q -k "fixture" 2>&1 | tail -15
