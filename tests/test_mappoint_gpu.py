"""GPU parity: MapPoint maintenance + Frame::isInFrustum vs the oracle restatement of
reference src/MapPoint.cc:345-521 and src/Frame.cc:509-565.  Index / level / flag fields bit-exact; float fields
bit-exact too (same operation order, -ffp-contract=off)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lists(rng, P, nmax):
    base = rng.integers(0, 256, (P, 32), dtype=np.uint8)
    out = []
    for p in range(P):
        n = int(rng.integers(0, nmax + 1)) if p % 7 else (0 if p % 14 == 0 else nmax)
        d = np.repeat(base[p][None], n, 0)
        flips = rng.random((n, 256)) < 0.08          # noisy views of one descriptor: many ties in the medians
        d = np.bitwise_xor(d, np.packbits(flips, axis=1))
        out.append(d)
    return out


@pytest.mark.parametrize("nmax", [1, 2, 9, 40, 128, 200])
def test_distinctive_descriptors(nmax):
    from object_slam_amd import MapPointBatch
    from oracle import oracle_py as O
    rng = np.random.default_rng(nmax)
    P = 301
    lists = _lists(rng, P, nmax)
    mp = MapPointBatch()
    best, desc = mp.ComputeDistinctiveDescriptors(lists)
    for p in range(P):
        ref = O.distinctive_descriptor(lists[p])
        assert best[p] == ref, (p, len(lists[p]))
        if ref >= 0:
            assert np.array_equal(desc[p], lists[p][ref])
        else:
            assert not desc[p].any()
    # identical observations: every median is 0 -> first index
    same = [np.repeat(lists[1][:1], 5, 0)] if len(lists[1]) else [np.zeros((5, 32), np.uint8)]
    assert mp.ComputeDistinctiveDescriptors(same)[0][0] == 0
    assert len(mp.ComputeDistinctiveDescriptors([])[0]) == 0


def test_update_normal_and_depth():
    from object_slam_amd import MapPointBatch
    from oracle import oracle_py as O
    rng = np.random.default_rng(5)
    P = 2000
    Pos = rng.normal(0, 5, (P, 3)).astype(np.float32)
    lists = [rng.normal(0, 3, (int(rng.integers(1, 30)), 3)).astype(np.float32) for _ in range(P)]
    OwRef = np.stack([l[0] for l in lists])
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    lsf = sf[rng.integers(0, 8, P)]
    out = MapPointBatch().UpdateNormalAndDepth(Pos, lists, OwRef, lsf, sf[-1])
    for p in range(P):
        ref = O.update_normal_depth(Pos[p], lists[p], OwRef[p], lsf[p], sf[-1])
        assert np.array_equal(out[p].view(np.uint32), ref.view(np.uint32)), p


@pytest.mark.parametrize("nmax", [1, 9, 40, 128])
def test_fused_small_update_matches_oracle(nmax):
    """oslam_mp_update_fused_device (the driver's small MapPoint updates in one launch: descriptors read from resident keyframe records through (record, keypoint)
    pairs) against the oracle's ComputeDistinctiveDescriptors and UpdateNormalAndDepth, bit for bit, incl. points without observations, a descriptor list shorter
    than the observation list, and the resident records."""
    import ctypes as C
    import torch
    from object_slam_amd._lib import lib, check
    from oracle import oracle_py as O
    L = lib()
    rng = np.random.default_rng(100 + nmax)
    P, nkf, nkp = 257, 12, 500
    kf_desc = [torch.from_numpy(rng.integers(0, 256, (nkp, 32), dtype=np.uint8)).cuda() for _ in range(nkf)]
    kf_host = [k.cpu().numpy() for k in kf_desc]
    recp = torch.tensor([k.data_ptr() for k in kf_desc], dtype=torch.int64, device="cuda")
    obs_start, desc_start, rec, lists, ow_lists = [0], [0], [], [], []
    for p in range(P):
        n = int(rng.integers(0, nmax + 1)) if p % 7 else (0 if p % 14 == 0 else nmax)
        nd = n if p % 5 else max(0, n - 2)          # some observations sit in culled keyframes: not in the descriptor list
        pairs = [(int(rng.integers(0, nkf)), int(rng.integers(0, nkp))) for _ in range(nd)]
        if nd >= 3 and p % 3 == 0:                   # near-duplicate views: ties in the medians
            pairs = [pairs[0]] * nd
        rec += pairs
        lists.append(np.stack([kf_host[r][k] for r, k in pairs]) if nd else np.zeros((0, 32), np.uint8))
        ow_lists.append(rng.normal(0, 3, (n, 3)).astype(np.float32))
        obs_start.append(obs_start[-1] + n); desc_start.append(desc_start[-1] + nd)
    Pos = rng.normal(0, 5, (P, 3)).astype(np.float32)
    OwRef = rng.normal(0, 3, (P, 3)).astype(np.float32)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    lsf = sf[rng.integers(0, 8, P)]
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dt))).cuda()
    d = dict(os=dev(obs_start, np.int32), ds=dev(desc_start, np.int32), rec=dev(np.array(rec, np.int32).reshape(-1, 2) if rec else np.zeros((1, 2), np.int32), np.int32),
             ow=dev(np.concatenate(ow_lists) if obs_start[-1] else np.zeros((1, 3), np.float32), np.float32), pos=dev(Pos, np.float32), ref=dev(OwRef, np.float32), lsf=dev(lsf, np.float32),
             items=dev(np.stack([np.zeros(P, np.int32), np.arange(P, dtype=np.int32)], 1), np.int32))
    tab = torch.full((P, 16), 7.0, dtype=torch.float32, device="cuda")
    tabp = torch.tensor([tab.data_ptr()], dtype=torch.int64, device="cuda")
    best = torch.full((P,), 99, dtype=torch.int32, device="cuda"); od = torch.full((P, 32), 9, dtype=torch.uint8, device="cuda"); o5 = torch.full((P, 5), 3.0, dtype=torch.float32, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    check(L.oslam_mp_update_fused_device(P, 1, 1, vp(d["os"]), vp(d["ds"]), vp(d["rec"]), vp(recp), vp(d["ow"]), vp(d["pos"]), vp(d["ref"]), vp(d["lsf"]), C.c_float(float(sf[-1])),
                                         vp(d["items"]), vp(tabp), vp(best), vp(od), vp(o5), None))
    torch.cuda.synchronize()
    best, od, o5, recs = best.cpu().numpy(), od.cpu().numpy(), o5.cpu().numpy(), tab.cpu().numpy()
    for p in range(P):
        ref = O.distinctive_descriptor(lists[p])
        assert best[p] == ref, (p, len(lists[p]), best[p], ref)
        assert np.array_equal(od[p], lists[p][ref]) if ref >= 0 else not od[p].any()
        n = obs_start[p + 1] - obs_start[p]
        assert np.array_equal(recs[p, :3], Pos[p])
        if n == 0:
            assert not o5[p].any() and (recs[p, 3:] == 7.0).all()
            continue
        r5 = O.update_normal_depth(Pos[p], ow_lists[p], OwRef[p], lsf[p], sf[-1])
        assert np.array_equal(o5[p].view(np.uint32), r5.view(np.uint32)), p
        assert np.array_equal(recs[p, 3:6], r5[:3]) and recs[p, 6] == r5[4] and recs[p, 7] == r5[3]
        if ref >= 0:
            assert np.array_equal(recs[p, 8:].view(np.uint8), lists[p][ref])
        else:
            assert (recs[p, 8:] == 7.0).all()


def test_update_normal_and_depth_from_local_ba_windows():
    """oslam_mp_update_windows_device (the MapPoint updates after a local BA, from the solved windows' own arrays: include/oslam_slam.h oslam_job_mp_window_t)
    against the oracle's UpdateNormalAndDepth over the surviving observations of every point — bit for bit — incl. erased edges, skipped points, several windows
    per call and the resident records."""
    import ctypes as C
    import torch
    from object_slam_amd._lib import lib, check
    from oracle import oracle_py as O
    L = lib()
    rng = np.random.default_rng(11)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    wins, P, E, Kt = [], 0, 0, 0
    for w in range(3):
        K, nP = int(rng.integers(3, 30)), int(rng.integers(50, 400))
        Ow = rng.normal(0, 3, (K, 3)).astype(np.float32)
        pos = rng.normal(0, 5, (nP, 3)).astype(np.float32)
        pstart, ekf, erase, ref, lsf, skip = [0], [], [], [], [], []
        for j in range(nP):
            kfs = np.sort(rng.choice(K, int(rng.integers(2, min(K, 12) + 1)), replace=False))
            er = (rng.random(len(kfs)) < 0.15).astype(np.uint8)
            if er.all():
                er[0] = 0
            live = kfs[er == 0]
            ekf += kfs.tolist(); erase += er.tolist(); pstart.append(len(ekf))
            ref.append(int(live[rng.integers(0, len(live))])); lsf.append(float(sf[rng.integers(0, 8)])); skip.append(int(rng.random() < 0.05))
        wins.append(dict(K=K, nP=nP, Ow=Ow, pos=pos, pstart=np.array(pstart, np.int32), ekf=np.array(ekf, np.int32), erase=np.array(erase, np.uint8), ref=np.array(ref, np.int32),
                         lsf=np.array(lsf, np.float32), skip=np.array(skip, np.uint8), p0=P, e0=E, k0=Kt))
        P += nP; E += len(ekf); Kt += K
    cat = lambda f: np.concatenate([f(w) for w in wins])
    e0 = cat(lambda w: w["e0"] + w["pstart"][:-1]).astype(np.int32)
    ne = cat(lambda w: np.diff(w["pstart"])).astype(np.int32)
    kbase = cat(lambda w: np.full(w["nP"], w["k0"])).astype(np.int32)
    items = np.stack([np.zeros(P, np.int32), np.arange(P, dtype=np.int32)], 1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tab = torch.full((P, 16), 7.0, dtype=torch.float32, device="cuda")
    tabp = torch.tensor([tab.data_ptr()], dtype=torch.int64, device="cuda")
    d = dict(items=dev(items), e0=dev(e0), ne=dev(ne), kb=dev(kbase), ref=dev(cat(lambda w: w["ref"])), lsf=dev(cat(lambda w: w["lsf"])), skip=dev(cat(lambda w: w["skip"])),
             pos=dev(cat(lambda w: w["pos"])), ekf=dev(cat(lambda w: w["ekf"])), er=dev(cat(lambda w: w["erase"])), Ow=dev(cat(lambda w: w["Ow"])))
    out = torch.zeros((P, 5), dtype=torch.float32, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    check(L.oslam_mp_update_windows_device(P, vp(d["items"]), vp(tabp), vp(d["e0"]), vp(d["ne"]), vp(d["kb"]), vp(d["ref"]), vp(d["lsf"]), vp(d["skip"]), vp(d["pos"]), vp(d["ekf"]),
                                           vp(d["er"]), vp(d["Ow"]), C.c_float(float(sf[-1])), vp(out), None))
    torch.cuda.synchronize()
    out, rec = out.cpu().numpy(), tab.cpu().numpy()
    for w in wins:
        for j in range(w["nP"]):
            g = w["p0"] + j
            assert np.array_equal(rec[g, :3], w["pos"][j])
            if w["skip"][j]:
                assert not out[g].any() and (rec[g, 3:8] == 7.0).all()
                continue
            sl = slice(w["pstart"][j], w["pstart"][j + 1])
            live = w["ekf"][sl][w["erase"][sl] == 0]
            ref = O.update_normal_depth(w["pos"][j], w["Ow"][live], w["Ow"][w["ref"][j]], w["lsf"][j], sf[-1])
            assert np.array_equal(out[g].view(np.uint32), ref.view(np.uint32)), (g, out[g], ref)
            assert np.array_equal(rec[g, 3:6], ref[:3]) and rec[g, 6] == ref[4] and rec[g, 7] == ref[3] and (rec[g, 8:] == 7.0).all()


@pytest.mark.parametrize("th", [1.0, 3.0])
def test_is_in_frustum(th):
    from object_slam_amd import MapPointBatch, ORBmatcher, QUERY_DTYPE
    from object_slam_amd.synth import KITTI_K
    from oracle import oracle_py as O
    rng = np.random.default_rng(11)
    M = 20000
    K5 = np.asarray(KITTI_K, np.float32)[:5]
    assert len(K5) == 5
    bounds = np.array([0, 0, 1241, 376], np.float32)
    # pose: small rotation + translation
    a = 0.1
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = R
    Tcw[:3, 3] = [0.3, -0.1, 0.5]
    Pw = np.concatenate([rng.uniform(-30, 30, (M, 2)), rng.uniform(-5, 60, (M, 1))], 1).astype(np.float32)
    Ow = -R.T @ Tcw[:3, 3]
    dirs = Pw - Ow
    dist = np.linalg.norm(dirs, axis=1)
    Pn = dirs / dist[:, None] + rng.normal(0, 0.6, (M, 3))
    Pn = (Pn / np.linalg.norm(Pn, axis=1)[:, None]).astype(np.float32)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    maxD = (dist * rng.uniform(0.5, 6.0, M)).astype(np.float32)
    # a block of points exactly on predicted-level boundaries: maxD = dist * 1.2^k
    maxD[:2000] = (dist[:2000].astype(np.float32) * sf[rng.integers(0, 8, 2000)])
    minD = (maxD / sf[-1]).astype(np.float32)
    obs = (rng.random(M) < 0.9).astype(np.uint8)
    desc = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    logsf = np.float32(np.log(np.float32(1.2)))
    args = (Pw, Pn, maxD, minD, obs, desc, Tcw, K5, bounds, 0.5, logsf, sf, th)
    got = MapPointBatch().isInFrustum(*args)
    ref = O.is_in_frustum(*args)
    assert 0.05 * M < int((ref["flags"] & 1).sum()) < 0.9 * M
    for f in QUERY_DTYPE.names:
        assert np.array_equal(got[f], ref[f]), f
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
    # the queries feed SearchByProjection(F, vpMapPoints) unchanged
    assert got.dtype == QUERY_DTYPE and ORBmatcher is not None
