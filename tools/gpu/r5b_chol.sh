# solver after the ISA fixes (factor_block inlined, branch-free loads, hoisted bases): phase stamps, LBA parity tests, kernel stats at 40 / 128 windows
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_chol; mkdir -p $O
OSLAM_LIB_PATH=tools/_build/liboslam_hip_prof.so timeout -k 10 120 python tools/chol_lds_phase_prof.py 27 | tee $O/phase27.txt
OSLAM_LIB_PATH=tools/_build/liboslam_hip_prof.so timeout -k 10 120 python tools/chol_lds_phase_prof.py 31 | tee $O/phase31.txt
timeout -k 10 500 python -m pytest tests/test_lba_gpu.py -x -q -m gpu > $O/test_lba.log 2>&1 || { tail -30 $O/test_lba.log; exit 1; }
tail -2 $O/test_lba.log
NB=40 MODES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt40 -o kt -- python3 tools/lba_win_prof.py > $O/kt40.log 2>&1
NB=128 MODES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt128 -o kt -- python3 tools/lba_win_prof.py > $O/kt128.log 2>&1
cat $O/kt40.log $O/kt128.log | grep "mode"
for d in kt40 kt128; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo $d; head -12 $f | cut -d, -f1-4 | cut -c1-150; done
