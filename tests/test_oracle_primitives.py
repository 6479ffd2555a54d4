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


# ---- exact known answers from the PUBLISHED definitions of the OpenCV 3.2 fixed-point paths: each expectation below is computed here, in
# integer / float32 arithmetic written from the library's documented algorithm, independently of oracle/*.cc; the oracle must reproduce it bit for bit ----
def _resize_linear_8u_by_definition(src, dw, dh):
    """cv::resize INTER_LINEAR, CV_8UC1 (imgproc/src/imgwarp.cpp, OpenCV 3.2): 11-bit coefficients (INTER_RESIZE_COEF_SCALE = 2048) computed in float and
    rounded to short; horizontal pass in int; vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    sh, sw = src.shape

    def coeffs(dn, sn):
        scale = np.float64(sn) / np.float64(dn)
        idx, ab = [], []
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(np.floor(f))
            f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= sn - 1:
                s, f = sn - 1, np.float32(0)
            a0 = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
            a1 = int(np.rint(np.float32(f * np.float32(2048))))
            idx.append(s); ab.append((a0, a1))
        return idx, ab
    xi, xa = coeffs(dw, sw)
    yi, ya = coeffs(dh, sh)
    rows = np.zeros((sh, dw), np.int64)
    for y in range(sh):
        for x in range(dw):
            s = xi[x]
            rows[y, x] = int(src[y, s]) * xa[x][0] + int(src[y, min(s + 1, sw - 1)]) * xa[x][1]
    out = np.zeros((dh, dw), np.uint8)
    for y in range(dh):
        s = yi[y]
        r0, r1 = rows[s], rows[min(s + 1, sh - 1)]
        b0, b1 = ya[y]
        for x in range(dw):
            out[y, x] = (((b0 * (int(r0[x]) >> 4)) >> 16) + ((b1 * (int(r1[x]) >> 4)) >> 16) + 2) >> 2
    return out


def test_resize_fixed_point_known_answers(oracle):
    # the hand-computable case: 4x4 -> 3x3 (scale 4/3: source positions 1/6, 1.5, 2 5/6 -> coefficients (1707, 341), (1024, 1024), (341, 1707))
    src = np.array([[0, 40, 80, 120], [10, 50, 90, 130], [200, 160, 120, 80], [255, 0, 255, 0]], np.uint8)
    want = _resize_linear_8u_by_definition(src, 3, 3)
    # first row by hand: y coefficients (1707, 341) over rows 0 / 1
    r0 = [0 * 1707 + 40 * 341, 40 * 1024 + 80 * 1024, 80 * 341 + 120 * 1707]
    r1 = [10 * 1707 + 50 * 341, 50 * 1024 + 90 * 1024, 90 * 341 + 130 * 1707]
    hand = [((((1707 * (a >> 4)) >> 16) + ((341 * (b >> 4)) >> 16) + 2) >> 2) for a, b in zip(r0, r1)]
    assert list(want[0]) == hand == [8, 62, 115]
    np.testing.assert_array_equal(oracle.resize_linear_u8(src, 3, 3), want)
    # the pyramid's own ratios (x 1/1.2 cascades) and an odd size, random content
    rng = np.random.default_rng(11)
    for (sw, sh, dw, dh) in [(64, 48, 53, 40), (53, 40, 44, 33), (37, 29, 31, 24), (1241 // 8, 47, 129, 39)]:
        s = rng.integers(0, 256, (sh, sw)).astype(np.uint8)
        np.testing.assert_array_equal(oracle.resize_linear_u8(s, dw, dh), _resize_linear_8u_by_definition(s, dw, dh))


GAUSS7_S2_TAPS = [18, 34, 49, 55, 49, 34, 18]   # cvRound(256 * getGaussianKernel(7, 2)): 8-bit fixed point, sum 257


def test_gaussian_7x7_sigma2_integer_weights_and_rounding(oracle):
    k = np.exp(-(np.arange(7) - 3.0) ** 2 / (2 * 2.0 ** 2))
    k = (k / k.sum()).astype(np.float32)   # getGaussianKernel(7, 2, CV_32F)
    assert [int(np.rint(float(v) * 256)) for v in k] == GAUSS7_S2_TAPS and sum(GAUSS7_S2_TAPS) == 257
    t = np.array(GAUSS7_S2_TAPS, np.int64)
    # impulse: the response IS the outer product of the taps, rounded once at the end ((v + 2^15) >> 16: FixedPtCastEx of the column filter)
    src = np.zeros((15, 15), np.uint8)
    src[7, 7] = 255
    want = np.zeros((15, 15), np.int64)
    want[4:11, 4:11] = (255 * np.outer(t, t) + 32768) >> 16
    for sse2 in (True, False):
        np.testing.assert_array_equal(oracle.gaussian_blur(src, sse2), want.astype(np.uint8))
    assert want[7, 7] == (255 * 55 * 55 + 32768) >> 16 == 12
    # vertical step 0 | 200 with BORDER_REFLECT_101: every row the same, columns = partial sums of the taps
    src = np.zeros((12, 16), np.uint8)
    src[:, 8:] = 200
    row = []
    for x in range(16):
        acc = 0
        for j in range(7):
            xx = x + j - 3
            xx = -xx if xx < 0 else (2 * 15 - xx if xx > 15 else xx)   # reflect 101
            acc += int(t[j]) * int(src[0, xx])
        row.append((257 * acc + 32768) >> 16)   # the row pass of a constant column profile: x 257 (sum of the taps)
    for sse2 in (True, False):
        out = oracle.gaussian_blur(src, sse2)
        assert all(list(out[y]) == row for y in range(12)), (list(out[0]), row)
    assert row[:5] == [0] * 5 and row[-5:] == [(257 * 257 * 200 + 32768) >> 16] * 5 == [202] * 5


def _fast_atan2_by_definition(y, x):
    """cv::fastAtan2 (core/src/mathfuncs_core.cpp, OpenCV 3.2 scalar path): 7th-order odd polynomial on [0, 1] in float32, then the octant reflections."""
    f = np.float32
    s = f(180.0 / np.pi)
    p1, p3, p5, p7 = f(0.9997878412794807) * s, f(-0.3258083974640975) * s, f(0.1555786518463281) * s, f(-0.04432655554792128) * s
    x, y = f(x), f(y)
    ax, ay = abs(x), abs(y)
    eps = f(2.220446049250313e-16)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f(180.0) - a
    if y < 0:
        a = f(360.0) - a
    return f(a)


def test_fast_atan2_polynomial_at_the_octant_boundaries(oracle):
    pts = [(0, 7), (7, 7), (7, 0), (7, -7), (0, -7), (-7, -7), (-7, 0), (-7, 7)]   # (y, x): 0, 45, 90, ... 315 degrees
    pts += [(1, 1000), (1000, 1), (999, 1000), (1000, 999), (-1, 1000), (-1000, -999), (3, -4), (-12345, 67890)]
    for (y, x) in pts:
        got = np.float32(oracle.fast_atan2(y, x))
        want = _fast_atan2_by_definition(y, x)
        assert got == want, (y, x, got, want)
    # the polynomial is not exact at 45 degrees: OpenCV 3.2 returns 44.99.. / 45.00.. there, the same in every octant by symmetry
    d45 = _fast_atan2_by_definition(7, 7)
    assert abs(float(d45) - 45.0) < 0.01
    assert np.float32(oracle.fast_atan2(7, -7)) == np.float32(180.0) - d45 and np.float32(oracle.fast_atan2(-7, 7)) == np.float32(360.0) - d45


def test_fast9_on_the_16_rotations_of_a_minimal_arc(oracle):
    """cv::FAST (TYPE_9_16): a pixel is a corner iff 9 CONTIGUOUS pixels of the 16-pixel Bresenham circle are all brighter than v + t or all darker than v - t —
    every start position of the arc, both polarities, and the 8-pixel arc that must NOT fire."""
    t = 20
    for start in range(16):
        for length, want in ((9, True), (8, False)):
            for bright in (True, False):
                img = np.full((9, 9), 100, np.uint8)
                for k in range(length):
                    dx, dy = CIRCLE[(start + k) % 16]
                    img[4 + dy, 4 + dx] = 100 + t + 1 if bright else 100 - t - 1
                got = {(int(q["x"]), int(q["y"])) for q in oracle.fast_9_16(img, t, nms=False)}
                assert ((4, 4) in got) == want, (start, length, bright, got)
                if want:   # the score of the centre = the largest threshold that still passes = t (the arc is exactly t + 1 away)
                    kn = [q for q in oracle.fast_9_16(img, t, nms=True) if (int(q["x"]), int(q["y"])) == (4, 4)]
                    assert len(kn) == 1 and int(kn[0]["response"]) == t
