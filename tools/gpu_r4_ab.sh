cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_LIB_PATH=$PWD/tools/_build/liboslam_hip_FP.so timeout -k 10 200 python tools/fast_phase_prof.py 2>&1 | tail -8
