set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r05_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r05_gpu_suite.log
O=gpurun_out/r5trace2
mkdir -p $O
export HIP_FORCE_DEV_KERNARG=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bt -- python3 bench.py --seqs 1024 --handles 2 --preroll 200 --no-extras --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err
echo "rc=$? (trace)"
head -12 $O/trace/bt_kernel_stats.csv | cut -d, -f1-5 | cut -c1-120
grep -n "copyBuffer\|k_copy_to_host" $O/trace/bt_kernel_stats.csv | cut -d, -f1-5
