set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_FUSECUR_CHECK=1 OSLAM_SLAM_CULL_CHECK=1 timeout -k 10 540 python tests/soak_s1.py 2 1000 1 sync > gpurun_out/r05_soak_sync.log 2>&1 || { tail -20 gpurun_out/r05_soak_sync.log; exit 1; }
tail -3 gpurun_out/r05_soak_sync.log | cut -c1-1500
OSLAM_SLAM_FUSECUR_CHECK=1 OSLAM_SLAM_CULL_CHECK=1 timeout -k 10 540 python tests/soak_s1.py 2 1000 1 deferred > gpurun_out/r05_soak_deferred.log 2>&1 || { tail -20 gpurun_out/r05_soak_deferred.log; exit 1; }
tail -3 gpurun_out/r05_soak_deferred.log | cut -c1-1500
