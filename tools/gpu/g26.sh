set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r05_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r05_gpu_suite.log
bash tools/gpu/soak.sh 2>&1 | grep -v "^oracle frame" | cut -c1-200
