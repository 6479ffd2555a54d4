set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r05_lba_tests_h.log 2>&1 || { tail -30 gpurun_out/r05_lba_tests_h.log; exit 1; }
tail -2 gpurun_out/r05_lba_tests_h.log
for L in 1 2 4 1 2 4; do echo lanes $L; OSLAM_LBA_UPD_LANES=$L NB=40,96,128,256 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows | sed 's/host to host.*window 0:.*->//'; done
