cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_CULL_STATS=1 python bench.py --no-extras --no-cpu-baseline --seqs 1024 --handles 2 > gpurun_out/r05_cullstats.json 2> gpurun_out/r05_cullstats.err
grep "cull stats" gpurun_out/r05_cullstats.err
