set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for AB in 1 2; do
OSLAM_EXTRA_FLAGS="-DOSLAM_SCHUR_ABLATE=$AB" python -m object_slam_amd.build -f > /dev/null 2>&1
cd /tmp && OSLAM_LBA_REC=1 NB=40 MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_c -o ab$AB -- python $R/tools/lba_win_prof.py > /dev/null 2>&1 || true
cd $R
echo "ABLATE=$AB"; grep "k_w_schur_rec\|k_w_lin\|k_w_update" gpurun_out/r05_prof_c/ab${AB}_kernel_stats.csv | cut -d, -f1-4
done
