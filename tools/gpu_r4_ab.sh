cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_poseopt_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/pose_sem_prof.py 1024 4 2>&1 | tail -5
OSLAM_LIB_PATH=$PWD/tools/_build/liboslam_hip_PP.so timeout -k 10 300 python tools/pose_sem_prof.py 256 4 2>&1 | tail -9
