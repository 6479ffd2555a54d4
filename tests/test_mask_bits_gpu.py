"""GPU parity of the one-bit-per-pixel mask path (oslam_mask_bits_device): the keypoint test of Frame::BuildObject2DsRGBD (reference src/Frame.cc:262-272)
from bitmaps equals the byte form and the CPU oracle on masks with values other than {0, 255}, odd widths, unaligned pitches and keypoints whose window
leaves the image; the semantic driver run is bit-identical with the bitmaps switched off."""
import ctypes as C

import numpy as np
import pytest

from object_slam_amd import _lib, slam
from object_slam_amd._lib import check

pytestmark = pytest.mark.gpu


def _masks(rng, n, H, W, pitch):
    m = np.zeros((n, H, pitch), np.uint8)
    for o in range(n):
        for _ in range(3):
            x0, y0 = rng.integers(0, W - 8), rng.integers(0, H - 8)
            w, h = rng.integers(8, W), rng.integers(8, H)
            m[o, y0:y0 + h, x0:x0 + w] = 255
        holes = rng.integers(0, W, (40, 2))
        for x, y in holes:
            m[o, min(y, H - 1), x] = rng.choice([0, 254, 1])       # only == 255 counts
    m[:, :, W:] = 255                                              # padding bytes must not leak into the bitmaps
    return m


@pytest.mark.parametrize("H,W,pitch", [(480, 640, 640), (37, 100, 104), (120, 131, 133), (64, 64, 64)])
def test_keypoint_test_from_bitmaps_matches_bytes_and_oracle(oracle, H, W, pitch):
    import torch
    rng = np.random.default_rng(H * 1000 + W)
    L = _lib.lib()
    B, nm, cap = 3, 3, 600
    masks = _masks(rng, B * nm, H, W, pitch)
    kp_dt = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])
    keys = np.zeros((B, cap), kp_dt)
    keys["x"] = rng.uniform(-2, W + 2, (B, cap)).astype(np.float32)
    keys["y"] = rng.uniform(-2, H + 2, (B, cap)).astype(np.float32)
    keys["x"][:, :50] = np.float32(W / 2) + rng.integers(-5, 5, (B, 50))       # integer coordinates
    keys["x"][:, 50:60] = np.nextafter(np.float32(32.0), np.float32(0))        # float sums that cross a power of two
    keys["x"][:, 60:80] = rng.uniform(9.0, 11.0, (B, 20)).astype(np.float32)   # windows that start around column 0
    n_kps = np.array([cap, cap - 7, 1], np.int32)
    d_masks = torch.from_numpy(masks).cuda()
    ptrs = np.array([d_masks[i].data_ptr() for i in range(B * nm)], np.uint64)
    t = lambda a: torch.from_numpy(a).cuda()
    d_ptrs, d_keys, d_n = t(ptrs.view(np.int64)), t(keys.view(np.uint8).reshape(B, -1)), t(n_kps)
    d_m0, d_nm = t(np.arange(B, dtype=np.int32) * nm), t(np.full(B, nm, np.int32))
    WB = (W + 63) // 64
    d_bits = torch.zeros(B * nm * H * WB, dtype=torch.int64, device="cuda")
    out_a = torch.zeros((B, cap), dtype=torch.uint8, device="cuda")
    out_b = torch.zeros((B, cap), dtype=torch.uint8, device="cuda")
    vp = lambda x: C.c_void_p(x.data_ptr())
    check(L.oslam_frame_object_kp_test_batch_device(vp(d_keys), cap, vp(d_n), B, vp(d_ptrs), vp(d_m0), vp(d_nm), H, W, pitch, vp(out_a), None))
    check(L.oslam_mask_bits_device(vp(d_ptrs), B * nm, H, W, pitch, vp(d_bits), None))
    check(L.oslam_frame_object_kp_test_bits_batch_device(vp(d_keys), cap, vp(d_n), B, vp(d_bits), vp(d_m0), vp(d_nm), H, W, vp(out_b), None))
    torch.cuda.synchronize()
    a, b = out_a.cpu().numpy(), out_b.cpu().numpy()
    bits = d_bits.cpu().numpy().view(np.uint64).reshape(B * nm, H, WB)
    ref = np.zeros((B * nm, H, WB * 64), bool)
    ref[:, :, :W] = masks[:, :, :W] == 255
    got = np.unpackbits(bits.view(np.uint8), axis=-1, bitorder="little").reshape(B * nm, H, WB * 64).astype(bool)
    assert np.array_equal(got, ref)
    for f in range(B):
        n = int(n_kps[f])
        assert np.array_equal(a[f, :n], b[f, :n]), f
        o = oracle.object_kp_test(keys[f, :n], np.ascontiguousarray(masks[f * nm:(f + 1) * nm, :, :W]))
        assert np.array_equal(b[f, :n], o), f
    if W == 640:
        assert (b[0, :cap] != 0).sum() > 10 and (b[0, :cap] == 0).sum() > 10      # both outcomes occur


def test_mask_bitmaps_leave_the_semantic_run_unchanged(monkeypatch):
    """Driver level: object_kps + pose_opt2 through the bitmaps (default) against the byte kernels (OSLAM_SLAM_NO_MASK_BITS): identical poses, statistics and
    semantic edge counts, with masks resident in HBM and with masks on the host."""
    import torch
    from object_slam_amd import scene
    W, H, n = 640, 480, 14
    q = scene.make_rgbd_sequence(3, n, speed=2.0)

    def run(on_device):
        sysm = slam.System(slam.make_config(W, H, 1))
        poses = []
        for t in range(n):
            if on_device:
                g = torch.from_numpy(q["gray"][t]).cuda(); d = torch.from_numpy(q["depth"][t]).cuda(); m = torch.from_numpy(q["masks"][t]).cuda()
                torch.cuda.synchronize()
                objs = [dict(masks=[m[o].data_ptr() for o in range(3)], track_ids=q["track_ids"])]
                T, st = sysm.TrackRGBD([g.data_ptr()], [d.data_ptr()], [t / 30.0], objects=objs, on_device=True, gray_stride=W, depth_pitch=W, mask_stride=W)
            else:
                objs = [dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])]
                T, st = sysm.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
            assert st[0] == slam.OK
            poses.append(T[0].copy())
        return np.array(poses), sysm.stats(0)

    pa, sa = run(True)
    ph, sh = run(False)
    monkeypatch.setenv("OSLAM_SLAM_NO_MASK_BITS", "1")
    pb, sb = run(True)
    assert sa["semantic_edges"] > 0 and sa == sb == sh
    assert np.array_equal(pa, pb) and np.array_equal(pa, ph)
