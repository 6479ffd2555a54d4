cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_CULL_CHECK=1 python tools/gpu/cullcheck.py 30 2>&1 | grep -v amdgpu.ids | tail -5
