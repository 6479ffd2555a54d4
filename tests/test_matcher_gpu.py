"""GPU parity: HIP windowed Hamming matcher (grid binning, SearchByProjection both variants,
sequential claim order, rotation histogram) vs the CPU oracle.  Bit-exact indices/distances."""
import numpy as np
import pytest

from object_slam_amd import KP_DTYPE, QUERY_DTYPE, ORBextractor, ORBmatcher, synth

pytestmark = pytest.mark.gpu

SCALE = np.array([1.2 ** i for i in range(8)], np.float32)


def _rand_frame(rng, N, w=640, h=480, clustered=False):
    k = np.zeros(N, KP_DTYPE)
    if clustered:   # few distinct positions: many queries compete for the same keypoints
        cx = rng.uniform(50, w - 50, 12)
        cy = rng.uniform(50, h - 50, 12)
        sel = rng.integers(0, 12, N)
        k["x"] = (cx[sel] + rng.normal(0, 4, N)).astype(np.float32)
        k["y"] = (cy[sel] + rng.normal(0, 4, N)).astype(np.float32)
    else:
        k["x"] = rng.uniform(-5, w + 5, N).astype(np.float32)   # a few outside the grid on purpose
        k["y"] = rng.uniform(-5, h + 5, N).astype(np.float32)
    k["octave"] = rng.integers(0, 8, N)
    k["angle"] = rng.uniform(0, 360, N).astype(np.float32)
    desc = rng.integers(0, 256, (N, 32)).astype(np.uint8)
    uR = np.where(rng.random(N) < 0.7, k["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    return k, uR, desc


def _rand_queries(rng, k, uR, desc, M, noise_bits, p_block=0.8, few_desc=False):
    q = np.zeros(M, QUERY_DTYPE)
    src = rng.integers(0, len(k), M)
    q["u"] = k["x"][src] + rng.normal(0, 3, M)
    q["v"] = k["y"][src] + rng.normal(0, 3, M)
    q["ur"] = q["u"] - (k["x"][src] - uR[src]) + rng.normal(0, 2, M)
    lvl = np.clip(k["octave"][src] + rng.integers(-1, 2, M), 0, 7)
    q["radius"] = (rng.choice([2.5, 4.0], M) * rng.choice([1, 3, 7], M) * SCALE[lvl]).astype(np.float32)
    q["minLevel"] = lvl - 1
    q["maxLevel"] = lvl
    q["flags"] = (rng.random(M) < 0.95).astype(np.int32) | ((rng.random(M) < p_block).astype(np.int32) << 1)
    q["angle"] = (k["angle"][src] + rng.normal(0, 20, M)) % 360
    d = desc[src].copy()
    if few_desc:   # force Hamming ties: only a handful of distinct descriptors
        d = desc[src % 5].copy()
    flip = rng.random((M, 256)) < noise_bits
    d ^= np.packbits(flip, axis=1, bitorder="little")
    q["desc"] = d
    return q


def _check(oracle, m, k, uR, desc, blocked, q, use_ratio, check_ori):
    bounds = (0.0, 0.0, 640.0, 480.0)
    nm, qm, qd, km = m.search_window(k, uR, desc, blocked, bounds, q, use_ratio, check_ori)
    onm, oqm, oqd, okm = oracle.search_by_projection(k, uR, desc, blocked, bounds, q, m.mfNNratio, use_ratio, check_ori)
    np.testing.assert_array_equal(qm, oqm)
    np.testing.assert_array_equal(qd, oqd)
    np.testing.assert_array_equal(km, okm)
    assert nm == onm
    return nm


@pytest.mark.parametrize("seed", range(6))
def test_search_window_random(oracle, seed):
    rng = np.random.default_rng(seed)
    m = ORBmatcher(0.8, True, max_keypoints=2400, max_queries=4096)
    N, M = [(1000, 1500), (2000, 4000), (300, 50), (1, 1), (2400, 4096), (800, 3000)][seed]
    k, uR, desc = _rand_frame(rng, N, clustered=(seed % 2 == 1))
    blocked = (rng.random(N) < 0.1).astype(np.uint8)
    q = _rand_queries(rng, k, uR, desc, M, 0.08, few_desc=(seed in (1, 5)))
    n1 = _check(oracle, m, k, uR, desc, blocked, q, True, False)
    n2 = _check(oracle, m, k, uR, desc, blocked, q, False, True)
    if N > 100:
        assert n1 > 0 and n2 > 0
    m.close()


def test_empty_and_edge(oracle):
    m = ORBmatcher(0.8, True, max_keypoints=512, max_queries=512)
    rng = np.random.default_rng(0)
    k, uR, desc = _rand_frame(rng, 100)
    q = _rand_queries(rng, k, uR, desc, 10, 0.05)
    # no queries / no keypoints
    nm, qm, qd, km = m.search_window(k, uR, desc, None, (0, 0, 640, 480), q[:0], True, False)
    assert nm == 0 and len(qm) == 0 and np.all(km == -1)
    nm, qm, qd, km = m.search_window(k[:0], uR[:0], desc[:0], None, (0, 0, 640, 480), q, True, False)
    assert nm == 0 and np.all(qm == -1)
    # monocular (uRight NULL) and no blocked array
    _ = m.search_window(k, None, desc, None, (0, 0, 640, 480), q, True, False)
    o = oracle.search_by_projection(k, None, desc, None, (0, 0, 640, 480), q, 0.8, True, False)
    np.testing.assert_array_equal(_[1], o[1])
    # capacity errors are loud
    from object_slam_amd import OslamError
    with pytest.raises(OslamError):
        m.search_window(np.zeros(600, KP_DTYPE), None, np.zeros((600, 32), np.uint8), None, (0, 0, 640, 480), q, True, False)
    m.close()


def test_last_frame_projection_on_extracted_frames(oracle):
    """SearchByProjection(Cur, Last) on real extractor output of two stream frames."""
    frames, offs = synth.make_stream(40, 640, 480, seed=9)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480)
    kl, dl = ex(frames[10])
    kc, dc = ex(frames[13])
    sf = ex.GetScaleFactors()
    fx = fy = 520.0
    cx, cy, bf, b = 320.0, 240.0, 40.0, 40.0 / 520.0
    Z0 = 2.0
    # fronto-parallel plane at depth Z0, cameras translate by the crop offset
    du, dv = (offs[13] - offs[10]).astype(np.float64)
    Xw = np.stack([(kl["x"] - cx) * Z0 / fx, (kl["y"] - cy) * Z0 / fy, np.full(len(kl), Z0)], 1).astype(np.float32)
    Tlw = np.eye(4, dtype=np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[0, 3] = -du * Z0 / fx
    Tcw[1, 3] = -dv * Z0 / fy
    rng = np.random.default_rng(1)
    has_mp = ((rng.random(len(kl)) < 0.9).astype(np.uint8)) | ((rng.random(len(kl)) < 0.7).astype(np.uint8) << 1)
    uRc = (kc["x"] - bf / Z0).astype(np.float32)
    bounds = (0.0, 0.0, 640.0, 480.0)
    for th, mono in ((15.0, False), (7.0, False), (15.0, True)):
        m = ORBmatcher(0.9, True, max_keypoints=1200, max_queries=1200)
        nm, qm, qd, km = m.search_last_frame(kc, uRc, dc, None, bounds, Xw, has_mp, kl, dl, Tcw, Tlw,
                                             (fx, fy, cx, cy, bf, b), sf, th, mono)
        oq = oracle.project_last_frame(Xw, has_mp, kl, dl, Tcw, Tlw, (fx, fy, cx, cy, bf, b), bounds, sf, th, mono)
        gq = m.debug_queries(len(kl), q_stride=1200)
        for f in ("u", "v", "ur", "radius", "minLevel", "maxLevel", "flags", "angle"):
            np.testing.assert_array_equal(gq[f], oq[f], err_msg=f)
        np.testing.assert_array_equal(gq["desc"][oq["flags"] & 1 == 1], oq["desc"][oq["flags"] & 1 == 1])
        onm, oqm, oqd, okm = oracle.search_by_projection(kc, uRc, dc, None, bounds, oq, 0.9, False, True)
        np.testing.assert_array_equal(qm, oqm)
        np.testing.assert_array_equal(qd, oqd)
        np.testing.assert_array_equal(km, okm)
        assert nm == onm
        assert nm > 300, nm   # the same scene 3 frames apart: most points re-found
        m.close()
    ex.close()


@pytest.mark.parametrize("seed", range(4))
def test_fuse_search(oracle, seed):
    """Search half of ORBmatcher::Fuse: chi2-gated best match <= TH_LOW."""
    rng = np.random.default_rng(100 + seed)
    N, M = [(1000, 800), (2000, 3000), (200, 60), (1500, 1500)][seed]
    k, uR, desc = _rand_frame(rng, N, clustered=(seed == 1))
    q = _rand_queries(rng, k, uR, desc, M, 0.05)
    q["radius"] = (3.0 * SCALE[np.clip(q["maxLevel"], 0, 7)]).astype(np.float32)
    inv = (1.0 / (SCALE * SCALE)).astype(np.float32)
    m = ORBmatcher(0.6, True, max_keypoints=2400, max_queries=4096)
    nf, qm, qd = m.fuse_search(k, uR, desc, (0.0, 0.0, 640.0, 480.0), q, inv)
    onf, oqm, oqd = oracle.fuse_search(k, uR, desc, (0.0, 0.0, 640.0, 480.0), q, inv)
    np.testing.assert_array_equal(qm, oqm)
    np.testing.assert_array_equal(qd, oqd)
    assert nf == onf and (nf > 10 or N < 300)
    m.close()


def _bow_pair(rng, N1, N2, n_nodes):
    k1, uR1, d1 = _rand_frame(rng, N1)
    src = rng.integers(0, N1, N2)
    k2 = np.zeros(N2, KP_DTYPE)
    k2["x"] = k1["x"][src] - rng.uniform(2, 30, N2)
    k2["y"] = k1["y"][src] + rng.normal(0, 1.0, N2)
    k2["octave"] = np.clip(k1["octave"][src] + rng.integers(-1, 2, N2), 0, 7)
    k2["angle"] = (k1["angle"][src] + rng.normal(0, 8, N2)) % 360
    d2 = d1[src].copy()
    d2 ^= np.packbits(rng.random((N2, 256)) < 0.06, axis=1, bitorder="little")
    node1 = rng.integers(0, n_nodes, N1).astype(np.uint32)
    node2 = node1[src].copy()
    wrong = rng.random(N2) < 0.15
    node2[wrong] = rng.integers(0, n_nodes, wrong.sum())
    uR2 = np.where(rng.random(N2) < 0.6, k2["x"] - rng.uniform(1, 30, N2), -1).astype(np.float32)
    return k1, uR1, d1, node1, k2, uR2, d2, node2


@pytest.mark.parametrize("seed", range(4))
def test_search_by_bow(oracle, seed):
    from object_slam_amd import BowMatcher
    rng = np.random.default_rng(200 + seed)
    N1, N2, nn = [(1000, 1000, 120), (2000, 2300, 40), (50, 80, 5), (1500, 900, 600)][seed]
    k1, uR1, d1, node1, k2, uR2, d2, node2 = _bow_pair(rng, N1, N2, nn)
    valid1 = (rng.random(N1) < 0.8).astype(np.uint8)
    bm = BowMatcher()
    for ratio, ori in ((0.7, True), (0.9, False)):
        nm, mf = bm.SearchByBoW(k1, d1, valid1, node1, k2, d2, node2, ratio, ori)
        onm, omf = oracle.search_by_bow(k1, d1, valid1, node1, k2, d2, node2, ratio, ori)
        np.testing.assert_array_equal(mf, omf)
        assert nm == onm and (nm > 5 or N1 < 100)
    bm.close()


@pytest.mark.parametrize("seed", range(3))
def test_search_for_triangulation(oracle, seed):
    from object_slam_amd import BowMatcher
    rng = np.random.default_rng(300 + seed)
    N1, N2, nn = [(1000, 1000, 100), (2000, 2200, 50), (300, 200, 10)][seed]
    k1, uR1, d1, node1, k2, uR2, d2, node2 = _bow_pair(rng, N1, N2, nn)
    mp1 = (rng.random(N1) < 0.4).astype(np.uint8)
    mp2 = (rng.random(N2) < 0.4).astype(np.uint8)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-7, (3, 3)).astype(np.float32)
    sigma2 = (SCALE * SCALE).astype(np.float32)
    bm = BowMatcher()
    for only_stereo in (False, True):
        nm, m12 = bm.SearchForTriangulation(k1, d1, uR1, mp1, node1, k2, d2, uR2, mp2, node2, F12, 700.0, 240.0, SCALE, sigma2, only_stereo, True)
        onm, om12 = oracle.search_for_triangulation(k1, d1, uR1, mp1, node1, k2, d2, uR2, mp2, node2, F12, 700.0, 240.0, SCALE, sigma2, only_stereo, True)
        np.testing.assert_array_equal(m12, om12)
        assert nm == onm and (nm > 5 or N1 < 400)
    bm.close()
