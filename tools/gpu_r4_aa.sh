cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export POSE_PROF_B=4096
for v in T128 T128ns T64 T64ns; do
  echo $v; OSLAM_LIB_PATH=$PWD/tools/_build/liboslam_hip_$v.so timeout -k 10 200 python tools/pose_prof.py 2>&1 | grep "batch" || exit 1
done
