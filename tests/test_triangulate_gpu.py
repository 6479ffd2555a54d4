"""GPU parity: LocalMapping::CreateNewMapPoints per-match core (reference src/LocalMapping.cc:291-432) vs the oracle.
Accept flags identical; positions <= 1e-5 relative (device hypot/atan2/cos vs glibc differ by <= 1 ulp inside the
Jacobi rotations and the stereo-parallax threshold)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SF = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
LS2 = (SF * SF).astype(np.float32)


def _pose(rng, t_scale):
    a = rng.normal(0, 0.08, 3)
    th = np.linalg.norm(a)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = R
    Tcw[:3, 3] = rng.normal(0, t_scale, 3)
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, :3] = Tcw[:3, :3].T
    Twc[:3, 3] = -(Tcw[:3, :3].T @ Tcw[:3, 3])
    return Tcw, Twc


def _make_kf(rng, X, Tcw, cam, stereo_frac, noise):
    from object_slam_amd import KP_DTYPE
    fx, fy, cx, cy, bf = cam
    Pc = X @ Tcw[:3, :3].T.astype(np.float64) + Tcw[:3, 3]
    z = Pc[:, 2]
    N = len(X)
    k = np.zeros(N, KP_DTYPE)
    octv = rng.integers(0, 8, N)
    sig = np.sqrt(LS2[octv])
    k["x"] = fx * Pc[:, 0] / z + cx + rng.normal(0, noise, N) * sig
    k["y"] = fy * Pc[:, 1] / z + cy + rng.normal(0, noise, N) * sig
    k["octave"] = octv
    k["size"] = 31 * SF[octv]
    st = (rng.random(N) < stereo_frac) & (z > 0.1)
    depth = np.where(st, z * (1 + rng.normal(0, 0.01, N)), -1).astype(np.float32)
    ur = np.where(st, k["x"] - bf / np.maximum(depth, 1e-3), -1).astype(np.float32)
    raw = k.copy()
    raw["x"] += 0.3   # mvKeys differs from mvKeysUn (distortion); only UnprojectStereo reads it
    return k, raw, ur, depth


@pytest.mark.parametrize("seed,stereo_frac,baseline", [(0, 0.0, 0.5), (1, 0.6, 0.5), (2, 0.6, 0.02), (3, 1.0, 1.5)])
def test_triangulate_matches(seed, stereo_frac, baseline):
    from object_slam_amd import MapPointBatch
    from object_slam_amd.mappoint import make_tri_kf
    from object_slam_amd.synth import KITTI_K
    from oracle import oracle_py as O
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = KITTI_K
    cam8 = np.array([fx, fy, cx, cy, np.float32(1) / np.float32(fx), np.float32(1) / np.float32(fy), bf, bf / fx], np.float32)
    N = 1500
    X = np.stack([rng.uniform(-15, 15, N), rng.uniform(-4, 4, N), rng.uniform(2, 60, N)], 1)
    X[:50, 2] = rng.uniform(-5, 0.5, 50)          # behind / very close: cheirality and parallax rejections
    T1, W1 = _pose(rng, 0.05)
    kf1_arr = _make_kf(rng, X, T1, KITTI_K, stereo_frac, 0.7)
    kf2s, arrs2, matches = [], [], []
    for p in range(4):
        T2, W2 = _pose(rng, baseline * (p + 1) / 2)
        a = _make_kf(rng, X, T2, KITTI_K, stereo_frac, 0.7)
        perm = rng.permutation(N).astype(np.int32)
        a = tuple(v[perm] for v in a)
        inv = np.argsort(perm).astype(np.int32)
        m = rng.choice(N, 600 if p else 0, replace=False).astype(np.int32)   # first pair: empty match list
        i2 = inv[m].copy()
        bad = rng.random(len(m)) < 0.15                 # wrong matches: reprojection / scale gates
        i2[bad] = rng.integers(0, N, int(bad.sum()))
        kf2s.append((T2, W2, a))
        matches.append((np.sort(m), i2[np.argsort(m)]))
    mp = MapPointBatch()
    k1 = make_tri_kf(T1, W1, cam8, *kf1_arr)
    k2 = [make_tri_kf(T2, W2, cam8, *a) for T2, W2, a in kf2s]
    ratioFactor = np.float32(1.5) * np.float32(1.2)
    ok, x = mp.triangulate(k1, k2, matches, SF, LS2, ratioFactor)
    ref_ok, ref_x = [], []
    for (T2, W2, a), (i1, i2) in zip(kf2s, matches):
        o, xx = O.triangulate((T1, W1, cam8) + kf1_arr, (T2, W2, cam8) + a, i1, i2, SF, LS2, ratioFactor)
        ref_ok.append(o)
        ref_x.append(xx)
    ref_ok, ref_x = np.concatenate(ref_ok), np.concatenate(ref_x)
    assert len(ok) == len(ref_ok) == 1800
    assert np.array_equal(ok, ref_ok)
    n = int(ok.sum())
    assert 50 < n < 1750, n
    sel = ok.astype(bool)
    rel = np.abs(x[sel] - ref_x[sel]).max(1) / np.maximum(np.abs(ref_x[sel]).max(1), 1e-3)
    assert rel.max() <= 1e-5, rel.max()
    assert not x[~sel].any()
    # accepted points really are near the synthetic landmarks when the match was correct
    print("accepted", n, "max rel", rel.max())


def test_triangulate_rejects_bad_indices():
    from object_slam_amd import MapPointBatch, OslamError
    from object_slam_amd.mappoint import make_tri_kf
    from object_slam_amd.synth import KITTI_K
    rng = np.random.default_rng(9)
    fx, fy, cx, cy, bf = KITTI_K
    cam8 = np.array([fx, fy, cx, cy, 1 / fx, 1 / fy, bf, bf / fx], np.float32)
    X = np.stack([rng.uniform(-5, 5, 20), rng.uniform(-2, 2, 20), rng.uniform(4, 20, 20)], 1)
    T1, W1 = _pose(rng, 0.05)
    T2, W2 = _pose(rng, 0.5)
    k1 = make_tri_kf(T1, W1, cam8, *_make_kf(rng, X, T1, KITTI_K, 0.5, 0.5))
    k2 = make_tri_kf(T2, W2, cam8, *_make_kf(rng, X, T2, KITTI_K, 0.5, 0.5))
    mp = MapPointBatch()
    with pytest.raises(OslamError):
        mp.triangulate(k1, [k2], [(np.array([0, 25]), np.array([0, 1]))], SF, LS2, 1.8)
    ok, x = mp.triangulate(k1, [], [], SF, LS2, 1.8)
    assert len(ok) == 0
