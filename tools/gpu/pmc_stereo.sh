set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5stereo_pmc
mkdir -p $O
cd /tmp && HIP_FORCE_DEV_KERNARG=0 timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p1 -o p1 -- python3 $R/bench.py --workload stereo --seqs 512 --handles 2 --no-extras --no-cpu-baseline --preroll 4 --steps 4 --warmup 1 > $O/b1.json 2> $O/b1.err || tail -3 $O/b1.err
cd $R
python tools/pmc_summary.py $O/summary.json $(find $O/p1 -name "*counter_collection.csv") | grep "k_stereo\|k_octree \|k_orient" | cut -c1-400
