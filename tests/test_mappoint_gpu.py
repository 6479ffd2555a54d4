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
