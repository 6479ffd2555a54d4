set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/octree_levels_prof.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout -k 10 600 python -m pytest tests/test_extractor_gpu.py -x -q 2>&1 | tail -2
