set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4full
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || (tail -40 $O/pytest_gpu.log; exit 1)
tail -3 $O/pytest_gpu.log
bash tools/gpu_r4_pmc.sh > $O/pmc.log 2>&1 || (tail -20 $O/pmc.log; exit 1)
tail -14 $O/pmc.log
