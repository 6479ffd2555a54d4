cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4x
OSLAM_EXTRA_FLAGS=-gline-tables-only python object_slam_amd/build.py -f > gpurun_out/r4x/build.log 2>&1 || tail -5 gpurun_out/r4x/build.log
timeout -k 10 500 python tools/host_prof.py run 8192 8 10 > gpurun_out/r4x/run.log 2>&1; tail -2 gpurun_out/r4x/run.log
python tools/host_prof.py report gpurun_out/host_prof.samples 400 > gpurun_out/r4x/report.txt 2>&1
head -75 gpurun_out/r4x/report.txt
python tools/host_prof_ranges.py gpurun_out/r4x/report.txt 45 > gpurun_out/r4x/ranges.txt 2>&1
