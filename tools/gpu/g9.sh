set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_LBA_PROFILE=1 python -m object_slam_amd.build -f > /dev/null 2>&1
OSLAM_LBA_PROFILE=1 python tools/chol_lds_phase_prof.py 27
OSLAM_LBA_PROFILE=1 python tools/chol_lds_phase_prof.py 31
