cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_mappoint_gpu.py tests/test_capi_cpu.py -x -q 2>&1 | tail -15
