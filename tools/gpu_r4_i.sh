cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T="tests/test_slam_driver_gpu.py::test_hip_driver_200_frames_with_masks_matches_oracle_driver"
for e in "A=1" "OSLAM_LBA_SOLVER=3" "OSLAM_LBA_NO_SERVICE=1" "OSLAM_LBA_SERVICE_THREADS=1"; do
  echo "== $e"
  env $e timeout -k 10 500 python -m pytest "$T" -q -k deferred 2>&1 | grep -E "passed|failed|Differing|!=" | head -8
done
