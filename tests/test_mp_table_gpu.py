"""GPU tests of the resident map-point record operators (include/oslam_hip.h): table write after a MapPoint update, the gathers that build pose /
SearchByProjection / SearchLocalPoints inputs from records, the Fuse projection gates against the driver's host form, and the vocabulary-tree node
assignment — each against a plain numpy restatement of its contract."""
import ctypes as C

import numpy as np
import pytest

from object_slam_amd import _lib
from object_slam_amd._lib import check

pytestmark = pytest.mark.gpu
KP = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])


def _setup(rng, S, R):
    import torch
    tabs = [torch.from_numpy(rng.integers(0, 256, (R, 64), dtype=np.uint8)).cuda() for _ in range(S)]
    ptrs = torch.from_numpy(np.array([t.data_ptr() for t in tabs], np.uint64).view(np.int64)).cuda()
    return tabs, ptrs


def test_table_write_and_gathers():
    import torch
    L = _lib.lib()
    rng = np.random.default_rng(3)
    S, R, P = 3, 500, 400
    tabs, ptrs = _setup(rng, S, R)
    before = [t.cpu().numpy().copy() for t in tabs]
    vp = lambda x: C.c_void_p(x.data_ptr())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # ---- oslam_mp_table_write_device
    items = np.stack([rng.integers(0, S, P), rng.permutation(R)[:P]], 1).astype(np.int32)      # distinct ids: every record is written at most once
    n_obs = rng.integers(0, 4, P)
    n_obs[:20] = 0                                                                             # culled points: only the position follows
    obs_start = np.concatenate([[0], np.cumsum(n_obs)]).astype(np.int32)
    n_desc = np.minimum(n_obs, rng.integers(0, 4, P))
    desc_start = np.concatenate([[0], np.cumsum(n_desc)]).astype(np.int32)
    Pos, out5 = rng.normal(size=(P, 3)).astype(np.float32), rng.normal(size=(P, 5)).astype(np.float32)
    out_desc = rng.integers(0, 256, (P, 32), dtype=np.uint8)
    keep = [t(items), t(obs_start), t(desc_start), t(Pos), t(out5), t(out_desc)]
    check(L.oslam_mp_table_write_device(P, vp(keep[0]), vp(ptrs), vp(keep[1]), vp(keep[2]), vp(keep[3]), vp(keep[4]), vp(keep[5]), 1, 1, None))
    torch.cuda.synchronize()
    want = [b.copy() for b in before]
    for i in range(P):
        rec = want[items[i, 0]][items[i, 1]]
        f = rec[:32].view(np.float32)
        f[0:3] = Pos[i]
        if n_obs[i] > 0:
            f[3:6] = out5[i, 0:3]; f[6] = out5[i, 4]; f[7] = out5[i, 3]       # normal, minimum distance, maximum distance
            if n_desc[i] > 0:
                rec[32:] = out_desc[i]
    for s in range(S):
        assert np.array_equal(tabs[s].cpu().numpy(), want[s]), s
    tab_np = want
    # ---- oslam_mp_table_gather_device (batch b reads the records of slot b)
    stride, n = 128, np.array([100, 0, 128], np.int32)
    ids = rng.integers(-1, R, (S, stride)).astype(np.int32)
    Xw = torch.zeros((S, stride, 3), dtype=torch.float32, device="cuda"); de = torch.zeros((S, stride, 32), dtype=torch.uint8, device="cuda")
    k2 = [t(n), t(ids)]
    check(L.oslam_mp_table_gather_device(S, stride, vp(k2[0]), vp(k2[1]), vp(ptrs), vp(Xw), vp(de), None))
    torch.cuda.synchronize()
    for b in range(S):
        for i in range(n[b]):
            r = tab_np[b][ids[b, i]] if ids[b, i] >= 0 else np.zeros(64, np.uint8)
            assert np.array_equal(Xw[b, i].cpu().numpy().view(np.uint8), r[:12]) and np.array_equal(de[b, i].cpu().numpy(), r[32:])
    # ---- oslam_mp_table_positions_device
    m = 300
    sl, pid = rng.integers(0, S, m).astype(np.int32), rng.integers(0, R, m).astype(np.int32)
    out = torch.zeros((m, 3), dtype=torch.float32, device="cuda")
    k3 = [t(sl), t(pid)]
    check(L.oslam_mp_table_positions_device(m, vp(k3[0]), vp(k3[1]), vp(ptrs), vp(out), None))
    torch.cuda.synchronize()
    ref = np.stack([tab_np[sl[i]][pid[i]][:12].view(np.float32) for i in range(m)])
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # ---- oslam_pose_inputs_gather_device
    cap, B = 96, 2
    keys = np.zeros((S, cap), KP); keys["x"] = rng.uniform(0, 640, (S, cap)); keys["y"] = rng.uniform(0, 480, (S, cap)); keys["octave"] = rng.integers(0, 8, (S, cap))
    uR = rng.uniform(-1, 600, (S, cap)).astype(np.float32)
    slots, nn = np.array([2, 0], np.int32), np.array([96, 50], np.int32)
    pid2 = rng.integers(-1, R, (B, cap)).astype(np.int32)
    inv = (1.0 / 1.44 ** np.arange(8)).astype(np.float32)
    o = [torch.zeros((B, cap, 3), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 3), dtype=torch.float32, device="cuda"),
         torch.zeros((B, cap), dtype=torch.float32, device="cuda"), torch.zeros((B, cap), dtype=torch.uint8, device="cuda")]
    k4 = [t(slots), t(nn), t(pid2), t(keys.view(np.uint8).reshape(S, -1)), t(uR)]
    check(L.oslam_pose_inputs_gather_device(B, cap, vp(k4[0]), vp(k4[1]), vp(k4[2]), vp(ptrs), vp(k4[3]), vp(k4[4]), cap, C.c_void_p(inv.ctypes.data), 8, vp(o[0]), vp(o[1]), vp(o[2]),
                                            vp(o[3]), None))
    torch.cuda.synchronize()
    for b in range(B):
        for i in range(nn[b]):
            s_ = slots[b]
            assert np.array_equal(o[1][b, i].cpu().numpy(), np.array([keys["x"][s_, i], keys["y"][s_, i], uR[s_, i]], np.float32))
            assert o[2][b, i].item() == inv[keys["octave"][s_, i]] and o[3][b, i].item() == int(pid2[b, i] >= 0)
            want_x = tab_np[s_][pid2[b, i]][:12].view(np.float32) if pid2[b, i] >= 0 else np.zeros(3, np.float32)
            assert np.array_equal(o[0][b, i].cpu().numpy().view(np.uint32), want_x.view(np.uint32))
    # ---- oslam_mp_table_local_gather_device: two jobs fill the rows of their slots in the [S][stride] layout of SearchLocalPoints
    lst, Ms, sl2 = 256, [200, 64], [2, 0]
    stage = np.zeros(4096, np.uint8)
    jobs = np.zeros(2, np.dtype([("slot", "i4"), ("M", "i4"), ("ids_off", "u4"), ("obs_off", "u4")]))
    lid, lob, off = [], [], 0
    for q in range(2):
        a, b_ = rng.integers(0, R, Ms[q]).astype(np.int32), rng.integers(0, 2, Ms[q]).astype(np.uint8)
        jobs[q] = (sl2[q], Ms[q], off, off + 1024)
        stage[off:off + 4 * Ms[q]] = a.view(np.uint8); stage[off + 1024:off + 1024 + Ms[q]] = b_
        lid.append(a); lob.append(b_); off += 2048
    dPw, dPn = torch.zeros((S, lst, 3), dtype=torch.float32, device="cuda"), torch.zeros((S, lst, 3), dtype=torch.float32, device="cuda")
    dMx, dMn = torch.zeros((S, lst), dtype=torch.float32, device="cuda"), torch.zeros((S, lst), dtype=torch.float32, device="cuda")
    dOb, dDe = torch.zeros((S, lst), dtype=torch.uint8, device="cuda"), torch.zeros((S, lst, 32), dtype=torch.uint8, device="cuda")
    k5 = [t(jobs.view(np.uint8)), t(stage)]
    check(L.oslam_mp_table_local_gather_device(2, max(Ms), vp(k5[0]), vp(k5[1]), vp(ptrs), lst, vp(dPw), vp(dPn), vp(dMx), vp(dMn), vp(dOb), vp(dDe), None))
    torch.cuda.synchronize()
    for q in range(2):
        s_ = sl2[q]
        for i in range(Ms[q]):
            r = tab_np[s_][lid[q][i]]
            f = r[:32].view(np.float32)
            assert np.array_equal(dPw[s_, i].cpu().numpy().view(np.uint32), f[0:3].view(np.uint32)) and np.array_equal(dPn[s_, i].cpu().numpy().view(np.uint32), f[3:6].view(np.uint32))
            assert dMn[s_, i].cpu().numpy().view(np.uint32) == f[6:7].view(np.uint32) and dMx[s_, i].cpu().numpy().view(np.uint32) == f[7:8].view(np.uint32)
            assert dOb[s_, i].item() == lob[q][i] and np.array_equal(dDe[s_, i].cpu().numpy(), r[32:])


def test_bow_nodes_match_the_tree_descent():
    import torch
    L = _lib.lib()
    rng = np.random.default_rng(5)
    top, sub = rng.integers(0, 2 ** 63, (10, 4), dtype=np.int64).view(np.uint64), rng.integers(0, 2 ** 63, (10, 10, 4), dtype=np.int64).view(np.uint64)
    n, cap = 3, 300
    counts = np.array([300, 17, 0], np.int32)
    desc = rng.integers(0, 256, (n, cap, 32), dtype=np.uint8)
    desc[0, :10] = np.frombuffer(top[3].tobytes(), np.uint8)          # exact hits and ties take the first minimum
    d = torch.from_numpy(desc).cuda()
    ptrs = torch.from_numpy(np.array([d[i].data_ptr() for i in range(n)], np.uint64).view(np.int64)).cuda()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    keep = [t(counts), t(top.view(np.int64)), t(sub.view(np.int64))]
    out = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    vp = lambda x: C.c_void_p(x.data_ptr())
    check(L.oslam_bow_nodes_device(vp(ptrs), vp(keep[0]), n, cap, vp(keep[1]), vp(keep[2]), vp(out), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    pc = lambda a, b: int(np.unpackbits((a ^ b).view(np.uint8)).sum())
    for i in range(n):
        for k in range(counts[i]):
            v = desc[i, k].view(np.uint64)
            b1 = int(np.argmin([pc(v, top[c]) for c in range(10)]))
            b2 = int(np.argmin([pc(v, sub[b1, c]) for c in range(10)]))
            assert got[i, k] == 11 + 10 * b1 + b2, (i, k)


def test_fuse_queries_match_the_host_form():
    """k_fuse_queries against the arithmetic of the driver's host form (ORBmatcher::Fuse's projection gates, reference src/ORBmatcher.cc:840-890), restated
    here in numpy float32 / float64 operator by operator."""
    import torch
    L = _lib.lib()
    rng = np.random.default_rng(9)
    f32, f64 = np.float32, np.float64
    S, R, n, stride = 2, 600, 2, 512
    K5 = np.array([535.4, 539.2, 320.1, 247.6, 40.0], f32)
    bounds = np.array([0.0, 0.0, 640.0, 480.0], f32)
    scale = (f32(1.2) ** np.arange(8)).astype(f32)
    logS = f32(np.log(f32(1.2)))
    recs = np.zeros((S, R, 16), f32)
    recs[:, :, 0:3] = rng.normal(0, 1.5, (S, R, 3)) + np.array([0, 0, 3.0])
    nrm = rng.normal(size=(S, R, 3)); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    recs[:, :, 3:6] = np.where(rng.random((S, R, 1)) < 0.7, recs[:, :, 0:3] / np.linalg.norm(recs[:, :, 0:3], axis=-1, keepdims=True), nrm)   # mean viewing direction: camera -> point
    recs[:, :, 6] = rng.uniform(0.3, 2.0, (S, R)); recs[:, :, 7] = recs[:, :, 6] * rng.uniform(2.0, 8.0, (S, R))
    recs = recs.astype(f32)
    raw = recs.view(np.uint8).reshape(S, R, 64).copy()
    raw[:, :, 32:] = rng.integers(0, 256, (S, R, 32), dtype=np.uint8)
    tabs = [torch.from_numpy(raw[s]).cuda() for s in range(S)]
    ptrs = torch.from_numpy(np.array([t_.data_ptr() for t_ in tabs], np.uint64).view(np.int64)).cuda()
    slots, M = np.array([1, 0], np.int32), np.array([500, 321], np.int32)
    ids = rng.integers(-1, R, (n, stride)).astype(np.int32)
    excl = (rng.random((n, stride)) < 0.2).astype(np.uint8)
    T = np.tile(np.eye(4, dtype=f32), (n, 1, 1)); T[0, :3, 3] = (0.1, -0.05, 0.2); T[1, 0, 3] = -0.3
    Ow = np.stack([-(T[b, :3, :3].T @ T[b, :3, 3]) for b in range(n)]).astype(f32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    keep = [t(slots), t(M), t(ids), t(excl), t(T.reshape(n, 16)), t(Ow)]
    qd = np.dtype([("u", "f4"), ("v", "f4"), ("ur", "f4"), ("radius", "f4"), ("minLevel", "i4"), ("maxLevel", "i4"), ("flags", "i4"), ("angle", "f4"), ("desc", "u1", 32)])
    out = torch.zeros((n, stride, 64), dtype=torch.uint8, device="cuda")
    vp = lambda x: C.c_void_p(x.data_ptr())
    cp = lambda a: C.c_void_p(a.ctypes.data)
    check(L.oslam_fuse_queries_device(n, stride, vp(keep[0]), vp(keep[1]), vp(keep[2]), vp(keep[3]), vp(ptrs), vp(keep[4]), vp(keep[5]), cp(K5), cp(bounds), C.c_float(3.0),
                                      C.c_float(float(logS)), cp(scale), 8, vp(out), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(qd).reshape(n, stride)
    active = near = 0
    for b in range(n):
        for i in range(M[b]):
            g = got[b, i]
            want_active = False
            if ids[b, i] >= 0 and not excl[b, i]:
                r = raw[slots[b], ids[b, i]]
                x = r[:32].view(f32)
                pc = []
                for k in range(3):
                    s_ = f32(T[b, k, 0] * x[0]); s_ = f32(s_ + f32(T[b, k, 1] * x[1])); s_ = f32(s_ + f32(T[b, k, 2] * x[2]))
                    pc.append(f32(f64(s_) + f64(T[b, k, 3])))
                if not pc[2] < 0:
                    invz = f32(f32(1) / pc[2])
                    u = f32(f32(K5[0] * f32(pc[0] * invz)) + K5[2]); v = f32(f32(K5[1] * f32(pc[1] * invz)) + K5[3])
                    if bounds[0] <= u < bounds[2] and bounds[1] <= v < bounds[3]:
                        PO = [f32(x[k] - Ow[b, k]) for k in range(3)]
                        dist = f32(np.sqrt(f64(PO[0]) * f64(PO[0]) + f64(PO[1]) * f64(PO[1]) + f64(PO[2]) * f64(PO[2])))
                        if not (dist < f32(f32(0.8) * x[6]) or dist > f32(f32(1.2) * x[7])):
                            dot = f64(PO[0]) * f64(x[3]) + f64(PO[1]) * f64(x[4]) + f64(PO[2]) * f64(x[5])
                            if not dot < 0.5 * f64(dist):
                                want_active = True
                                quo = f32(f32(np.log(f64(f32(x[7] / dist)))) / logS)
                                lvl = min(max(int(np.ceil(quo)), 0), 7)
                                if abs(quo - round(float(quo))) < 1e-5:      # the level sits on a rounding boundary of log(): either neighbour is acceptable
                                    near += 1
                                    assert abs(int(g["maxLevel"]) - lvl) <= 1
                                else:
                                    assert g["maxLevel"] == lvl and g["minLevel"] == lvl - 1 and g["radius"] == f32(f32(3.0) * scale[lvl]), (b, i)
                                assert g["u"] == u and g["v"] == v and g["ur"] == f32(u - f32(K5[4] * invz)), (b, i)
                                assert np.array_equal(g["desc"], r[32:])
            assert (g["flags"] == 1) == want_active, (b, i)
            active += want_active
    assert active > 100 and near < 5


def test_fuse_search_on_resident_grids_equals_queries_plus_window_search():
    """oslam_kf_grid_build_device + oslam_fuse_search_device (one launch, one candidate per thread, the keyframe's grid built once) against (a) a numpy restatement
    of Frame::AssignFeaturesToGrid's cell order (reference src/Frame.cc:455-470) for the grid arrays and (b) the staged path it replaces in the driver:
    oslam_fuse_queries_device followed by the LDS window search of ORBmatcher::Fuse (oslam_match_fuse_search, itself oracle-checked in tests/test_matcher_gpu.py) —
    index-exact."""
    import torch
    from object_slam_amd import ORBmatcher
    from object_slam_amd._lib import KP_DTYPE
    L = _lib.lib()
    rng = np.random.default_rng(31)
    f32 = np.float32
    S, R, n, stride, cap = 3, 900, 3, 1024, 1200
    K5 = np.array([535.4, 539.2, 320.1, 247.6, 40.0], f32)
    bounds = np.array([0.0, 0.0, 640.0, 480.0], f32)
    scale = (f32(1.2) ** np.arange(8)).astype(f32)
    inv_sigma2 = (f32(1) / (scale * scale)).astype(f32)
    logS = f32(np.log(f32(1.2)))
    # keyframes: clustered keypoints (several per cell, some outside the grid's rounding range), mixed octaves, 30 % without depth
    Ns = np.array([1100, 700, 0], np.int32)
    slots_kf = np.array([2, 0, 1], np.int32)          # the frame of job i sits in slot slots_kf[i] of the batch arrays
    keys = np.zeros((S, cap), KP_DTYPE); uR = np.full((S, cap), -1, f32); desc = rng.integers(0, 256, (S, cap, 32), dtype=np.uint8)
    base_desc = rng.integers(0, 256, (R, 32), dtype=np.uint8)
    for i in range(n):
        sl, N = slots_kf[i], Ns[i]
        keys["x"][sl, :N] = np.clip(rng.normal(320, 150, N), 0, 639.9).astype(f32); keys["y"][sl, :N] = np.clip(rng.normal(240, 110, N), 0, 479.9).astype(f32)
        keys["x"][sl, :N][:8] = 639.9; keys["y"][sl, :N][:8] = 479.9       # rounds to cell (64, 48): outside the grid
        keys["octave"][sl, :N] = rng.integers(0, 8, N); keys["angle"][sl, :N] = rng.uniform(0, 360, N)
        has_d = rng.random(N) < 0.7
        uR[sl, :N] = np.where(has_d, keys["x"][sl, :N] - rng.uniform(2, 30, N).astype(f32), -1).astype(f32)
        pick = rng.integers(0, R, N)                                         # descriptors near those of the map points, so that matches exist
        d = base_desc[pick].copy(); flip = rng.random((N, 32)) < 0.08; d[flip] ^= rng.integers(1, 256, (N, 32), dtype=np.uint8)[flip]
        desc[sl, :N] = d
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_keys, d_uR, d_desc, d_cnt = t(keys.view(np.uint8).reshape(S, -1)), t(uR), t(desc), t(np.array([Ns[1], Ns[2], Ns[0]], np.int32))   # counts by slot
    cell_end = torch.zeros((n, 3072), dtype=torch.int16, device="cuda"); cand = torch.zeros((n, cap, 4), dtype=torch.float32, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    gj = np.zeros((n, 5), np.int64)                                          # oslam_kf_grid_job_t: four pointers, int32 slot, int32 pad = 40 bytes
    for i in range(n):
        sl = int(slots_kf[i])
        gj[i] = (d_keys.data_ptr() + sl * cap * 28, d_uR.data_ptr() + sl * cap * 4, cell_end.data_ptr() + i * 3072 * 2, cand.data_ptr() + i * cap * 16, sl)
    d_gj = t(gj)
    vp = lambda x: C.c_void_p(x.data_ptr())
    cp = lambda a: C.c_void_p(a.ctypes.data)
    check(L.oslam_kf_grid_build_device(n, vp(d_gj), vp(d_cnt), cp(bounds), cap, vp(status), None))
    torch.cuda.synchronize()
    assert int(status[0]) == 0
    ce, cd = cell_end.cpu().numpy().view(np.uint16), cand.cpu().numpy()
    for i in range(n):
        sl, N = slots_kf[i], Ns[i]
        px = np.rint((keys["x"][sl, :N] - bounds[0]) * f32(64.0 / 640.0)).astype(int); py = np.rint((keys["y"][sl, :N] - bounds[1]) * f32(48.0 / 480.0)).astype(int)
        inside = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
        order = [k for c in range(3072) for k in np.nonzero(inside & (px * 48 + py == c))[0]]     # cells in (x, y) nesting, index order inside a cell
        counts = np.bincount((px * 48 + py)[inside], minlength=3072)
        assert np.array_equal(ce[i], np.cumsum(counts).astype(np.uint16))
        got = cd[i, :len(order)]
        assert np.array_equal(got[:, 0], keys["x"][sl][order]) and np.array_equal(got[:, 1], keys["y"][sl][order]) and np.array_equal(got[:, 2], uR[sl][order])
        assert np.array_equal(got[:, 3].view(np.uint32), (keys["octave"][sl][order].astype(np.uint32) << 16) | np.array(order, np.uint32))
        if N: assert 0 < len(order) <= N - 8      # (points clipped to the right / bottom edge round to column 64 / row 48: outside, as in the reference)
    # map-point records in front of the cameras, descriptors = base_desc
    recs = np.zeros((S, R, 16), f32)
    recs[:, :, 0:3] = rng.normal(0, 1.2, (S, R, 3)) + np.array([0, 0, 3.5])
    recs[:, :, 3:6] = recs[:, :, 0:3] / np.linalg.norm(recs[:, :, 0:3], axis=-1, keepdims=True)
    recs[:, :, 6] = rng.uniform(0.3, 2.0, (S, R)); recs[:, :, 7] = recs[:, :, 6] * rng.uniform(2.0, 8.0, (S, R))
    raw = recs.astype(f32).view(np.uint8).reshape(S, R, 64).copy()
    raw[:, :, 32:] = base_desc[None]
    # put the keypoints of the first two keyframes where some of the points project, so that the window search has candidates
    T = np.tile(np.eye(4, dtype=f32), (n, 1, 1)); T[0, :3, 3] = (0.1, -0.05, 0.2); T[1, 0, 3] = -0.3
    Ow = np.stack([-(T[b, :3, :3].T @ T[b, :3, 3]) for b in range(n)]).astype(f32)
    tabs = [torch.from_numpy(raw[s]).cuda() for s in range(S)]
    ptrs = t(np.array([t_.data_ptr() for t_ in tabs], np.uint64).view(np.int64))
    slots_mp, M = np.array([1, 0, 2], np.int32), np.array([900, 640, 333], np.int32)
    # keypoints where the map points of the job's sequence project (position within ~1.5 px, predicted level or the one below, descriptor a few bits away), so
    # that the window search has candidates that pass every gate; then the grids are built again
    for b in range(2):
        sl, N = slots_kf[b], Ns[b]
        X = recs[slots_mp[b]].astype(np.float64)
        pc = X[:, :3] @ T[b, :3, :3].T.astype(np.float64) + T[b, :3, 3]
        u = K5[0] * pc[:, 0] / pc[:, 2] + K5[2]; v = K5[1] * pc[:, 1] / pc[:, 2] + K5[3]
        dist = np.linalg.norm(X[:, :3] - Ow[b], axis=1)
        lvl = np.clip(np.ceil(np.log(X[:, 7] / dist) / float(logS)), 0, 7).astype(int)
        ok = np.nonzero((pc[:, 2] > 0) & (u > 5) & (u < 630) & (v > 5) & (v < 470))[0]
        take = ok[:min(len(ok), N - 50)]
        kk = np.arange(len(take)) + 20
        keys["x"][sl, kk] = (u[take] + rng.uniform(-1.5, 1.5, len(take))).astype(f32); keys["y"][sl, kk] = (v[take] + rng.uniform(-1.5, 1.5, len(take))).astype(f32)
        keys["octave"][sl, kk] = np.maximum(lvl[take] - rng.integers(0, 2, len(take)), 0)
        uR[sl, kk] = np.where(rng.random(len(take)) < 0.7, u[take] - K5[4] / pc[take, 2] + rng.uniform(-1, 1, len(take)), -1).astype(f32)
        d = base_desc[take].copy(); flip = rng.random(d.shape) < 0.05; d[flip] ^= (1 << rng.integers(0, 8, d.shape)).astype(np.uint8)[flip]
        desc[sl, kk] = d
    d_keys, d_uR, d_desc = t(keys.view(np.uint8).reshape(S, -1)), t(uR), t(desc)
    for i in range(n):
        sl = int(slots_kf[i])
        gj[i, 0], gj[i, 1] = d_keys.data_ptr() + sl * cap * 28, d_uR.data_ptr() + sl * cap * 4
    d_gj = t(gj)
    check(L.oslam_kf_grid_build_device(n, vp(d_gj), vp(d_cnt), cp(bounds), cap, vp(status), None))
    torch.cuda.synchronize()
    assert int(status[0]) == 0
    ids = rng.integers(-1, R, (n, stride)).astype(np.int32)
    excl = (rng.random((n, stride)) < 0.15).astype(np.uint8)
    keep = [t(slots_mp), t(M), t(ids), t(excl), t(T.reshape(n, 16)), t(Ow)]
    refs = np.zeros((n, 3), np.int64)
    for i in range(n):
        refs[i] = (cell_end.data_ptr() + i * 3072 * 2, cand.data_ptr() + i * cap * 16, d_desc.data_ptr() + int(slots_kf[i]) * cap * 32)
    d_refs = t(refs)
    qm = torch.full((n, stride), -7, dtype=torch.int32, device="cuda")
    for th in (3.0, 6.0):
        check(L.oslam_fuse_search_device(n, stride, vp(d_refs), vp(keep[0]), vp(keep[1]), vp(keep[2]), vp(keep[3]), vp(ptrs), vp(keep[4]), vp(keep[5]), cp(K5), cp(bounds),
                                         C.c_float(th), C.c_float(float(logS)), cp(scale), cp(inv_sigma2), 8, vp(qm), None))
        qbuf = torch.zeros((n, stride, 64), dtype=torch.uint8, device="cuda")
        check(L.oslam_fuse_queries_device(n, stride, vp(keep[0]), vp(keep[1]), vp(keep[2]), vp(keep[3]), vp(ptrs), vp(keep[4]), vp(keep[5]), cp(K5), cp(bounds), C.c_float(th),
                                          C.c_float(float(logS)), cp(scale), 8, vp(qbuf), None))
        torch.cuda.synchronize()
        got = qm.cpu().numpy()
        from object_slam_amd.matcher import QUERY_DTYPE
        queries = qbuf.cpu().numpy().view(QUERY_DTYPE).reshape(n, stride)
        m = ORBmatcher(max_keypoints=cap, max_queries=stride)
        fused = 0
        for b in range(n):
            sl, N = slots_kf[b], Ns[b]
            if N == 0:
                assert np.all(got[b, :M[b]] == -1)
                continue
            _, want, _ = m.fuse_search(keys[sl, :N], uR[sl, :N], desc[sl, :N], bounds, queries[b, :M[b]], inv_sigma2)
            assert np.array_equal(got[b, :M[b]], want), (th, b, np.nonzero(got[b, :M[b]] != want)[0][:10])
            fused += int((want >= 0).sum())
        m.close()
        assert fused > 60, fused
