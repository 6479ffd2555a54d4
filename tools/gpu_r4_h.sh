set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
for sch in 0 1; do
  echo "SCHUR=$sch"
  SCHUR=$sch MODES=1 NB=40,120,160,256 python3 tools/lba_win_prof.py 2>&1 | grep mode | cut -c1-120
done
echo "mode 2"
MODES=2 NB=40,120,160,256 python3 tools/lba_win_prof.py 2>&1 | grep mode | cut -c1-120
echo "stereo-like windows (10 KF, 7.5k edges): schur 0/1, mode 2"
SCHUR=0 MODES=1 NB=50,128 python3 tools/lba_win_prof.py 10 0 900 5 2>&1 | grep mode | cut -c1-150
SCHUR=1 MODES=1 NB=50,128 python3 tools/lba_win_prof.py 10 0 900 5 2>&1 | grep mode | cut -c1-150
MODES=2 NB=50,128 python3 tools/lba_win_prof.py 10 0 900 5 2>&1 | grep mode | cut -c1-150
