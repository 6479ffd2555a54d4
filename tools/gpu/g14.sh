set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_stereo_gpu.py tests/test_slam_driver_gpu.py -x -q -k "stereo" 2>&1 | tail -4
O=$R/gpurun_out/r5stereo2
mkdir -p $O
cd /tmp && HIP_FORCE_DEV_KERNARG=0 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stereo -- python3 $R/bench.py --workload stereo --seqs 512 --handles 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; }
cd $R
head -12 $O/stereo_kernel_stats.csv | sed 's/(oslam::[^"]*"/"/' | cut -c1-120
tail -1 $O/bench.json | cut -c1-200
