"""GPU parity: Frame::UndistortKeyPoints / ComputeImageBounds / ComputeStereoFromRGBD (reference src/Frame.cc:644-704,
:883-904) vs the oracle restatement of OpenCV 3.2's cvUndistortPoints.  Bit-exact (fp64 inside, same operation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TUM1_K = (517.306408, 516.469215, 318.643040, 255.313989)                      # reference Examples/RGB-D/TUM1.yaml:8-11
TUM1_D = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)                  # k1, k2, p1, p2, k3 (:13-17)
TUM2_K = (520.908620, 521.007327, 325.141442, 249.701764)
TUM2_D = (0.231222, -0.784899, -0.003257, -0.000105, 0.917205)


def _keys(rng, n, w, h):
    from object_slam_amd import KP_DTYPE
    k = np.zeros(n, KP_DTYPE)
    k["x"] = rng.uniform(0, w - 1, n).astype(np.float32)
    k["y"] = rng.uniform(0, h - 1, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["angle"] = rng.uniform(0, 360, n)
    k["size"] = 31
    k["response"] = rng.integers(7, 200, n)
    k["class_id"] = -1
    return k


@pytest.mark.parametrize("K,D", [(TUM1_K, TUM1_D), (TUM2_K, TUM2_D), (TUM2_K, TUM2_D[:4]), (TUM2_K, (0.0, 0.1, 0, 0)), (TUM2_K, ())])
def test_undistort_and_bounds(K, D):
    from object_slam_amd import FrameOps
    from oracle import oracle_py as O
    rng = np.random.default_rng(len(D))
    keys = _keys(rng, 5000, 640, 480)
    keys["x"][:4] = [0, 639, 0, 639]
    keys["y"][:4] = [0, 0, 479, 479]
    f = FrameOps()
    got = f.UndistortKeyPoints(keys, K, D)
    ref = O.undistort_keypoints(keys, K, D)
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
    if len(D) and D[0] != 0:
        assert np.abs(got["x"] - keys["x"]).max() > 1.0          # the distortion really moves points
        assert np.array_equal(got["octave"], keys["octave"]) and np.array_equal(got["angle"], keys["angle"])
    else:
        assert np.array_equal(got.view(np.uint8), keys.view(np.uint8))
    b = f.ComputeImageBounds(640, 480, K, D)
    rb = O.image_bounds(640, 480, K, D)
    assert np.array_equal(b.view(np.uint32), rb.view(np.uint32)), (b, rb)
    assert len(f.UndistortKeyPoints(keys[:0], K, D)) == 0


def test_stereo_from_rgbd():
    from object_slam_amd import FrameOps, OslamError
    from oracle import oracle_py as O
    rng = np.random.default_rng(4)
    keys = _keys(rng, 3000, 640, 480)
    f = FrameOps()
    un = f.UndistortKeyPoints(keys, TUM1_K, TUM1_D)
    depth = rng.uniform(0.3, 8.0, (480, 640)).astype(np.float32)
    depth[rng.random((480, 640)) < 0.2] = 0.0      # invalid depth
    depth[rng.random((480, 640)) < 0.05] = -1.0
    ur, dp = f.ComputeStereoFromRGBD(keys, un, depth, 40.0)
    rur, rdp = O.stereo_from_rgbd(keys, un, depth, 40.0)
    assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), rdp.view(np.uint32))
    assert 0.15 < (dp < 0).mean() < 0.35
    bad = keys.copy()
    bad["x"][7] = 700.0
    with pytest.raises(OslamError):
        f.ComputeStereoFromRGBD(bad, un, depth, 40.0)


def test_undistort_batch_device_matches_host_path():
    import ctypes as C
    import torch
    from object_slam_amd import FrameOps, KP_DTYPE, _lib
    rng = np.random.default_rng(8)
    B, S = 3, 1200
    counts = np.array([1200, 0, 777], np.int32)
    keys = np.stack([_keys(rng, S, 640, 480) for _ in range(B)])
    d_in = torch.from_numpy(keys.view(np.uint8).reshape(B, S * 28).copy()).cuda()
    d_out = torch.zeros_like(d_in)
    d_cnt = torch.from_numpy(counts).cuda()
    L = _lib.lib()
    k4 = (C.c_float * 4)(*TUM1_K)
    d5 = (C.c_float * 5)(*TUM1_D)
    _lib.check(L.oslam_frame_undistort_batch_device(C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), C.c_void_p(d_cnt.data_ptr()), 0, S, B, k4, d5, 5,
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().reshape(B, S * 28).view(KP_DTYPE).reshape(B, S)
    f = FrameOps()
    for b in range(B):
        ref = f.UndistortKeyPoints(keys[b][:counts[b]], TUM1_K, TUM1_D)
        assert np.array_equal(out[b][:counts[b]].view(np.uint8), ref.view(np.uint8))
        assert not out[b][counts[b]:].view(np.uint8).any()          # slots past the count are untouched
