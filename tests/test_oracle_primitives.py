"""CPU: pin the oracle's restated third-party primitives against independent definitions and the
known-answer tables of SURVEY.md (the reference itself has no tests or golden vectors)."""
import hashlib
import math

import numpy as np

import octree_model

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _segment_test(img, x, y, t):
    """FAST-9/16 by definition: >= 9 contiguous circle pixels all brighter than v+t or darker than v-t."""
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    for sign in (1, -1):
        flags = [(p > v + t) if sign > 0 else (p < v - t) for p in ring]
        run = best = 0
        for f in flags + flags:
            run = run + 1 if f else 0
            best = max(best, run)
        if best >= 9:
            return True
    return False


def _score_by_definition(img, x, y):
    """Largest t for which the pixel is still a corner (OpenCV cornerScore semantics)."""
    t = -1
    while t < 255 and _segment_test(img, x, y, t + 1):
        t += 1
    return t


def test_fast_matches_segment_test_definition(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 44)).astype(np.uint8)
    img[10:25, 12:30] = 200   # a real blob with corners
    img[15:20, 5:40] = 30
    for thr in (7, 20):
        kps = oracle.fast_9_16(img, thr, nms=False)
        got = {(int(k["x"]), int(k["y"])) for k in kps}
        want = {(x, y) for y in range(3, 37) for x in range(3, 41) if _segment_test(img, x, y, thr)}
        assert got == want
        # response (only stored with nms=True) = largest threshold still passing
        kn = oracle.fast_9_16(img, thr, nms=True)
        assert len(kn) > 0
        for k in kn:
            assert int(k["response"]) == _score_by_definition(img, int(k["x"]), int(k["y"]))
        # NMS = strict maximum over the 8 neighbours' scores (non-corners count 0), row-major order
        sc = np.zeros(img.shape, np.int32)
        for (x, y) in want:
            sc[y, x] = _score_by_definition(img, x, y)
        keep = [(x, y) for y in range(3, 37) for x in range(3, 41)
                if (x, y) in want and all(sc[y, x] > sc[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy)]
        assert [(int(k["x"]), int(k["y"])) for k in kn] == keep


def test_resize_close_to_float_bilinear(oracle):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    dst = oracle.resize_linear_u8(src, 533, 400)
    sx, sy = 640 / 533, 480 / 400
    xs = np.clip((np.arange(533) + 0.5) * sx - 0.5, 0, 639)
    ys = np.clip((np.arange(400) + 0.5) * sy - 0.5, 0, 479)
    x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, 639); fx = xs - x0
    y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, 479); fy = ys - y0
    s = src.astype(np.float64)
    ref = ((s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx) * (1 - fy)[:, None]
           + (s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx) * fy[:, None])
    assert np.abs(dst.astype(np.float64) - ref).max() <= 1.0   # fixed-point vs exact: within 1 LSB
    # identity resize is exact
    np.testing.assert_array_equal(oracle.resize_linear_u8(src, 640, 480), src)


def test_gaussian_close_to_float_and_rounding_variants(oracle):
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (61, 83)).astype(np.uint8)
    k = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    k /= k.sum()
    ref = correlate1d(correlate1d(src.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    for sse2 in (True, False):
        out = oracle.gaussian_blur(src, sse2)
        # 8-bit taps sum to 257/256 per pass in OpenCV 3.2: slight gain, stays within 3 grey levels
        assert np.abs(out.astype(np.float64) - ref * (257 / 256) ** 2).max() <= 1.0
    a, b = oracle.gaussian_blur(src, True), oracle.gaussian_blur(src, False)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1   # variants differ only at exact .5 ties
    np.testing.assert_array_equal(a[:, 80:], b[:, 80:])        # scalar tail columns x >= (w & ~3)
    flat = np.full((20, 20), 77, np.uint8)
    assert np.all(oracle.gaussian_blur(flat, True) == 78)      # 77 * (257/256)^2 = 77.6 -> 78


def test_fast_atan2(oracle):
    rng = np.random.default_rng(3)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        if x == 0 and y == 0:
            continue
        a = oracle.fast_atan2(y, x)
        r = math.degrees(math.atan2(y, x)) % 360
        d = abs(a - r)
        assert min(d, 360 - d) < 0.02, (y, x, a, r)
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, 5) == 0.0
    assert abs(oracle.fast_atan2(5, 0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0, -5) - 180.0) < 1e-4


def test_brief_pattern_and_tables(oracle):
    p = oracle.brief_pattern()
    digest = hashlib.sha256(",".join(str(int(v)) for v in p).encode()).hexdigest()
    # sha256 of the comma-joined reference table
    assert digest == "88df8ca875cc8db56799edd57bb914edad8acb2d48c202b7a464a575b55dbdb8"
    assert p.min() == -13 and p.max() == 12
    e = oracle.OrbExtractor(1000, 1.2, 8, 20, 7)
    t = e.tables()
    # SURVEY.md §8(a) A-1 known answers
    assert list(t["nfeatures_per_level"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    e2 = oracle.OrbExtractor(2000, 1.2, 8, 20, 7)
    assert list(e2.tables()["nfeatures_per_level"]) == [434, 362, 302, 251, 209, 175, 145, 122]
    # Appendix C level sizes
    e.extract(np.zeros((480, 640), np.uint8))
    assert [e.level_size(l) for l in range(8)] == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231),
                                                    (257, 193), (214, 161), (179, 134)]
    e2.extract(np.zeros((376, 1241), np.uint8))
    assert [e2.level_size(l) for l in range(8)] == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181),
                                                     (499, 151), (416, 126), (346, 105)]


def test_descriptor_distance_known_answers(oracle):
    rng = np.random.default_rng(4)
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, o) == 256
    for _ in range(200):
        a = rng.integers(0, 256, 32).astype(np.uint8)
        b = rng.integers(0, 256, 32).astype(np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())


def test_octree_array_form_equals_list_form(oracle):
    """The HIP kernel's array-in-list-order formulation (modelled in octree_model.py) against the
    oracle's literal std::list restatement of DistributeOctTree, incl. response and size ties."""
    e = oracle.OrbExtractor(2000)
    rng = np.random.default_rng(5)
    for trial in range(60):
        rw, rh = int(rng.integers(40, 1300)), int(rng.integers(40, 500))
        if round(np.float32(rw) / np.float32(rh)) < 1:
            continue
        n = int(rng.integers(1, 2500))
        idx = np.sort(rng.choice((rw - 6) * (rh - 6), size=min(n, (rw - 6) * (rh - 6)), replace=False))
        ys, xs = (idx // (rw - 6) + 3).astype(int), (idx % (rw - 6) + 3).astype(int)
        resp = rng.integers(7, 40, size=len(xs))
        N = int(rng.integers(1, 500))
        cand = np.zeros(len(xs), oracle.KP_DTYPE)
        cand["x"], cand["y"], cand["response"] = xs, ys, resp
        ref = e.distribute_octree(cand, 16, 16 + rw, 16, 16 + rh, N)
        got = octree_model.distribute(list(xs), list(ys), list(resp), rw, rh, N)
        assert len(ref) == len(got)
        assert np.array_equal(ref["x"], xs[got]) and np.array_equal(ref["y"], ys[got])


def test_grid_area_query(oracle):
    rng = np.random.default_rng(6)
    k = np.zeros(500, oracle.KP_DTYPE)
    k["x"] = rng.uniform(0, 640, 500)
    k["y"] = rng.uniform(0, 480, 500)
    k["octave"] = rng.integers(0, 8, 500)
    for _ in range(50):
        x, y, r = rng.uniform(0, 640), rng.uniform(0, 480), rng.uniform(1, 60)
        lo, hi = int(rng.integers(-1, 6)), int(rng.integers(-1, 8))
        got = set(oracle.features_in_area(k, (0, 0, 640, 480), x, y, r, lo, hi).tolist())
        check = (lo > 0) or (hi >= 0)
        want = set()
        for i in range(500):
            if check and (k["octave"][i] < lo or (hi >= 0 and k["octave"][i] > hi)):
                continue
            if abs(np.float32(k["x"][i]) - np.float32(x)) < np.float32(r) and abs(np.float32(k["y"][i]) - np.float32(y)) < np.float32(r):
                want.add(i)
        # the grid only prunes: every in-window keypoint inside the image grid must be returned
        assert got <= want
        # (a keypoint whose rounded cell lies just outside the scanned cell range can be missed:
        # that is the reference's behaviour, src/Frame.cc:572-584 vs :624-625)
        assert len(want - got) <= 2
