cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4trace
mkdir -p $O
export HIP_FORCE_DEV_KERNARG=0
for lm in deferred sync; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/$lm -o bt -- python3 bench.py --lm $lm --seqs 1024 --handles 2 --preroll 200 --no-extras --no-cpu-baseline > $O/bench_$lm.json 2> $O/bench_$lm.err
  echo "rc=$? ($lm)"
  tail -2 $O/bench_$lm.err | cut -c1-200
  DB=$(find $O/$lm -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_kernel_stats.py $DB > $O/kernel_stats_$lm.csv; head -12 $O/kernel_stats_$lm.csv | cut -c1-140; rm -f $DB; fi
done
