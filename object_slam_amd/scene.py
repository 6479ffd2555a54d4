"""Seeded synthetic 3-D scenes seen by a moving camera (S1 "TUM-shaped RGB-D" with 3 object masks, S3 "KITTI-shaped
stereo").  Textured planar surfaces (walls, the faces of boxes) are rendered through a per-plane homography evaluated in FIXED POINT
(int64 linear forms, floor division, 8-bit bilinear weights, integer mip selection), so the bytes of a frame do not depend on the
host's floating-point library; only the 3x3 homography coefficients are computed in float64 and rounded once.

What the scenes exercise that the fronto-parallel crops of synth.py cannot: rotation (<= 0.5 deg/frame) and forward/backward motion, so
keypoints migrate across pyramid levels and the forward/backward level band of ORBmatcher::SearchByProjection fires (reference
src/ORBmatcher.cc:1348-1349,1385-1390); depth varies over the image (RGB-D 1-4.5 m, stereo 5-60 m); local BA runs on a non-planar map;
three box objects give {0,255} instance masks for ObjectOptimizer::PoseOptimization2 (reference src/ObjectOptimizer.cc:624).
"""
import numpy as np

from .synth import make_canvas

DEPTH_FACTOR = 5000            # TUM depth PNG unit (reference Examples/RGB-D/TUM2.yaml DepthMapFactor: 5000)
TUM_K = (520.908620, 521.007327, 325.141442, 249.701764)      # reference Examples/RGB-D/TUM2.yaml:8-11
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)              # reference Examples/Stereo/KITTI00-02.yaml:8-11
KITTI_BASELINE = 386.1448 / 718.856                            # bf / fx = 0.5372 m (KITTI00-02.yaml:20)


def _rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


class Plane:
    """A textured rectangle P0 + a*U + b*V, 0 <= a < w, 0 <= b < h (metres); `label` 0 = background, k > 0 = object k."""

    def __init__(self, P0, U, V, w, h, density, seed, label=0):
        self.P0, self.U, self.V = np.asarray(P0, float), np.asarray(U, float), np.asarray(V, float)
        self.w, self.h, self.density, self.label = float(w), float(h), float(density), label
        tw, th = max(8, int(round(w * density))), max(8, int(round(h * density)))
        t0 = make_canvas(tw, th, seed, n_rects=max(8, tw * th // 350)).astype(np.int64)
        self.mips = [t0]
        for _ in range(3):       # 2x2 box filter, integer rounding
            t = self.mips[-1]
            hh, ww = t.shape[0] // 2 * 2, t.shape[1] // 2 * 2
            if hh < 2 or ww < 2:
                self.mips.append(t)
                continue
            t = t[:hh, :ww]
            self.mips.append((t[0::2, 0::2] + t[0::2, 1::2] + t[1::2, 0::2] + t[1::2, 1::2] + 2) >> 2)
        self.n = np.cross(self.U, self.V)


def facing(P0, U, V, w, h, density, seed, label, eye):
    """The rectangle oriented so that U x V points towards `eye` (mirrors U if it does not)."""
    P0, U, V = np.asarray(P0, float), np.asarray(U, float), np.asarray(V, float)
    if np.cross(U, V) @ (np.asarray(eye, float) - (P0 + U * w / 2 + V * h / 2)) < 0:
        P0, U = P0 + U * w, -U
    return Plane(P0, U, V, w, h, density, seed, label)


def box_planes(center, half, yaw, density, seed, label):
    """The six faces of a box (outward normals); U x V points outwards so back faces can be skipped."""
    R = _rot_xyz(0.0, yaw, 0.0)
    ax = [R[:, 0], R[:, 1], R[:, 2]]
    c = np.asarray(center, float)
    out = []
    k = 0
    for d in range(3):
        for sgn in (-1.0, 1.0):
            n = sgn * ax[d]
            a1, a2 = ax[(d + 1) % 3], ax[(d + 2) % 3]
            U, V = (a1, a2) if sgn > 0 else (a2, a1)          # U x V = +-(a1 x a2) = +-ax[d]
            h1, h2 = (half[(d + 1) % 3], half[(d + 2) % 3]) if sgn > 0 else (half[(d + 2) % 3], half[(d + 1) % 3])
            P0 = c + n * half[d] - U * h1 - V * h2
            out.append(Plane(P0, U, V, 2 * h1, 2 * h2, density, seed * 16 + k, label))
            k += 1
    return out


def _linear_forms(pl, Twc, K):
    """float64 coefficients (of u, v, 1) of A, B, N and the constant k0: s = density*A/N, t = density*B/N, z = k0/N."""
    fx, fy, cx, cy = K
    Rwc, C = Twc[:3, :3], Twc[:3, 3]
    M = Rwc @ np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1.0]])     # dw = M [u v 1]^T
    nN = pl.n @ M
    k0 = pl.n @ (pl.P0 - C)
    cU, cV = pl.U @ (C - pl.P0), pl.V @ (C - pl.P0)
    A = (cU * nN + k0 * (pl.U @ M)) * pl.density
    B = (cV * nN + k0 * (pl.V @ M)) * pl.density
    return A, B, nN, k0


def render(planes, Twc, K, W, H):
    """Returns (gray uint8 [H,W], depth uint16 [H,W] in 1/DEPTH_FACTOR m (0 = nothing hit / too far), label uint8 [H,W])."""
    gray = np.full((H, W), 30, np.uint8)
    zbuf = np.full((H, W), np.iinfo(np.int64).max, np.int64)
    label = np.zeros((H, W), np.uint8)
    Rcw = Twc[:3, :3].T
    C = Twc[:3, 3]
    fx, fy, cx, cy = K
    for pl in planes:
        if pl.n @ (C - pl.P0) <= 1e-9:       # back face (U x V points to the visible side)
            continue
        # bounding box of the projected quad (whole image if a corner is behind the camera)
        cor = np.array([pl.P0, pl.P0 + pl.U * pl.w, pl.P0 + pl.V * pl.h, pl.P0 + pl.U * pl.w + pl.V * pl.h])
        pc = (cor - C) @ Rcw.T
        if (pc[:, 2] > 0.05).all():
            uu = fx * pc[:, 0] / pc[:, 2] + cx
            vv = fy * pc[:, 1] / pc[:, 2] + cy
            u0, u1 = int(np.floor(uu.min())) - 1, int(np.ceil(uu.max())) + 2
            v0, v1 = int(np.floor(vv.min())) - 1, int(np.ceil(vv.max())) + 2
        elif (pc[:, 2] <= 0.05).all():
            continue
        else:
            u0, u1, v0, v1 = 0, W, 0, H
        u0, u1, v0, v1 = max(u0, 0), min(u1, W), max(v0, 0), min(v1, H)
        if u0 >= u1 or v0 >= v1:
            continue
        A, B, N, k0 = _linear_forms(pl, Twc, K)
        q = 2.0 ** 26 / max(np.abs(A).max(), np.abs(B).max(), np.abs(N).max(), abs(k0), 1e-30)
        sg = 1 if k0 > 0 else -1             # z = k0 / N > 0 where visible: make N positive there
        Ai, Bi, Ni = (np.rint(x * q * sg).astype(np.int64) for x in (A, B, N))
        K0 = int(np.rint(k0 * q * sg))
        # one extra row / column for the footprint differences
        us = np.arange(u0, u1 + 1, dtype=np.int64)[None, :]
        vs = np.arange(v0, v1 + 1, dtype=np.int64)[:, None]
        Nn = Ni[0] * us + Ni[1] * vs + Ni[2]
        ok = Nn > 0
        Nn = np.where(ok, Nn, 1)
        s = (256 * (Ai[0] * us + Ai[1] * vs + Ai[2])) // Nn
        t = (256 * (Bi[0] * us + Bi[1] * vs + Bi[2])) // Nn
        z = (DEPTH_FACTOR * K0) // Nn
        foot = np.maximum(np.maximum(np.abs(s[:-1, 1:] - s[:-1, :-1]), np.abs(t[:-1, 1:] - t[:-1, :-1])),
                          np.maximum(np.abs(s[1:, :-1] - s[:-1, :-1]), np.abs(t[1:, :-1] - t[:-1, :-1])))
        s, t, z, ok = s[:-1, :-1], t[:-1, :-1], z[:-1, :-1], ok[:-1, :-1]
        tw, th = pl.mips[0].shape[1], pl.mips[0].shape[0]
        inside = ok & (s >= 0) & (t >= 0) & (s < (tw - 1) * 256) & (t < (th - 1) * 256) & (z > 0)
        sub = zbuf[v0:v1, u0:u1]
        win = inside & (z < sub)
        if not win.any():
            continue
        lvl = (foot >= 384).astype(np.int64) + (foot >= 768) + (foot >= 1536)
        val = np.zeros(s.shape, np.int64)
        for L in range(4):
            m = win & (lvl == L)
            if not m.any():
                continue
            T = pl.mips[L]
            sl = np.maximum((s[m] - 128 * ((1 << L) - 1)) >> L, 0)
            tl = np.maximum((t[m] - 128 * ((1 << L) - 1)) >> L, 0)
            i, f = sl >> 8, sl & 255
            j, g = tl >> 8, tl & 255
            i0, j0 = np.minimum(i, T.shape[1] - 1), np.minimum(j, T.shape[0] - 1)
            i1, j1 = np.minimum(i + 1, T.shape[1] - 1), np.minimum(j + 1, T.shape[0] - 1)
            val[m] = ((256 - f) * (256 - g) * T[j0, i0] + f * (256 - g) * T[j0, i1] + (256 - f) * g * T[j1, i0] + f * g * T[j1, i1] + 32768) >> 16
        sub[win] = z[win]
        gray[v0:v1, u0:u1][win] = val[win].astype(np.uint8)
        label[v0:v1, u0:u1][win] = pl.label
    depth = np.where(zbuf > 65535, 0, zbuf).astype(np.uint16)     # beyond 13.1 m the 16-bit depth image has no value
    return gray, depth, label


# ---------------------------------------------------------------------------------------------------------------------------------
# S1: TUM-shaped RGB-D room with three box objects
# ---------------------------------------------------------------------------------------------------------------------------------
def make_tum_scene(seed):
    rng = np.random.Generator(np.random.PCG64(0x0B5E55ED + 7919 * seed))
    planes = []
    # back wall, yawed so that its depth runs from ~2.6 m to ~4.6 m across the image; a side wall closes the right-hand side
    yaw = np.deg2rad(rng.uniform(14, 22)) * (1 if seed % 2 == 0 else -1)
    R = _rot_xyz(0, yaw, 0)
    U, V = R[:, 0], np.array([0.0, 1.0, 0.0])
    centre = np.array([0.0, 0.0, 3.6])
    planes.append(facing(centre - U * 6.0 - V * 3.5, U, V, 12.0, 7.0, 150.0, 1000 + seed, 0, (0, 0, 0)))
    side = 1 if yaw > 0 else -1           # the wall end that comes towards the camera gets a side wall behind it
    R2 = _rot_xyz(0, yaw - side * np.deg2rad(70), 0)
    U2 = R2[:, 0]
    c2 = centre + U * (side * 5.0) + np.array([0, 0, 0.4])
    planes.append(facing(c2 - U2 * 4.0 - V * 3.5, U2, V, 8.0, 7.0, 150.0, 2000 + seed, 0, (0, 0, 0)))
    # three boxes in front of the wall, 1.3 - 2.4 m from the start position
    xs = [-0.75, 0.05, 0.85]
    for o in range(3):
        half = np.array([rng.uniform(0.22, 0.30), rng.uniform(0.20, 0.28), rng.uniform(0.12, 0.20)])
        c = np.array([xs[o] + rng.uniform(-0.08, 0.08), rng.uniform(-0.15, 0.25), rng.uniform(1.5, 2.3)])
        planes += box_planes(c, half, np.deg2rad(rng.uniform(-25, 25)), 380.0, 3000 + 10 * seed + o, o + 1)
    return planes


def tum_path(n, seed, speed=1.0):
    """Smooth SE3 path: <= 2 cm and <= 0.5 deg per frame at speed 1.  Returns Twc [n,4,4] float64."""
    rng = np.random.Generator(np.random.PCG64(0x5EED + 104729 * seed))
    ph = rng.uniform(0, 2 * np.pi, 6)
    t = np.arange(n) * speed
    x = 0.28 * np.sin(2 * np.pi * t / 200 + ph[0])
    y = 0.10 * np.sin(2 * np.pi * t / 150 + ph[1])
    z = 0.45 * np.sin(2 * np.pi * t / 400 + ph[2]) - 0.1
    yaw = np.deg2rad(6.0) * np.sin(2 * np.pi * t / 180 + ph[3])
    pitch = np.deg2rad(3.0) * np.sin(2 * np.pi * t / 130 + ph[4])
    roll = np.deg2rad(4.0) * np.sin(2 * np.pi * t / 220 + ph[5])
    T = np.tile(np.eye(4), (n, 1, 1))
    for i in range(n):
        T[i, :3, :3] = _rot_xyz(pitch[i], yaw[i], roll[i])
        T[i, :3, 3] = (x[i], y[i], z[i])
    return T


def depth_to_metres(depth_u16):
    """cv::Mat::convertTo(CV_32F, mDepthMapFactor) with mDepthMapFactor = 1.0f / 5000 (reference src/Tracking.cc:98-102,262)."""
    f = np.float64(np.float32(1.0) / np.float32(DEPTH_FACTOR))
    return (depth_u16.astype(np.float64) * f).astype(np.float32)


def make_rgbd_sequence(seed, n, width=640, height=480, speed=1.0, with_masks=True, first=0, count=None):
    """S1: dict(gray [n,H,W] u8, depth [n,H,W] f32 metres, masks [n,3,H,W] u8 {0,255}, track_ids [3], labels [3], Twc [n,4,4] f64
    relative to the first camera).  first / count: render only frames [first, first + count) of the n-frame stream (the images of a frame do
    not depend on the others, so a long stream can be rendered in pieces; Twc stays relative to frame 0 of the whole stream)."""
    planes = make_tum_scene(seed)
    Twc = tum_path(n, seed, speed)
    m = n - first if count is None else count
    gray = np.empty((m, height, width), np.uint8)
    depth = np.empty((m, height, width), np.float32)
    masks = np.zeros((m, 3, height, width), np.uint8) if with_masks else None
    for i in range(m):
        g, d, lab = render(planes, Twc[first + i], TUM_K, width, height)
        gray[i], depth[i] = g, depth_to_metres(d)
        if with_masks:
            for o in range(3):
                masks[i, o] = (lab == o + 1) * np.uint8(255)
    T0inv = np.linalg.inv(Twc[0])
    return dict(gray=gray, depth=depth, masks=masks, track_ids=np.array([0, 1, 2], np.int32), labels=np.array([56, 62, 73], np.int32),
                Twc=np.array([T0inv @ T for T in Twc[first:first + m]]))


# ---------------------------------------------------------------------------------------------------------------------------------
# S3: KITTI-shaped rectified stereo "street"
# ---------------------------------------------------------------------------------------------------------------------------------
def make_kitti_scene(seed, length=400.0):
    """A street along +z: facades on both sides, each turned 35-55 deg towards the oncoming camera, and a low-frequency backdrop
    beyond the end of the street.  Camera y points down; the facades span y in [-4.5, 1.5] m."""
    rng = np.random.Generator(np.random.PCG64(0xC177 + 15485863 * seed))
    planes = []
    V = np.array([0.0, 1.0, 0.0])
    z = 8.0
    k = 0
    while z < length:
        for side in (-1, 1):
            ang = np.deg2rad(rng.uniform(35, 55))
            U = np.array([-np.sin(ang), 0.0, side * np.cos(ang)])        # normal (-side cos, 0, -sin): towards the road centre and the camera
            w = rng.uniform(7.0, 12.0)
            c = np.array([side * rng.uniform(6.0, 9.0), -1.5, z + rng.uniform(0.0, 4.0)])
            planes.append(facing(c - U * w / 2 - V * 3.0, U, V, w, 6.0, 40.0, 500 + 97 * seed + k, 0, (0.0, 0.0, c[2] - 30.0)))
            k += 1
        z += rng.uniform(8.0, 12.0)
    U = np.array([1.0, 0.0, 0.0])
    zz = length + 60.0
    planes.append(facing(np.array([-150.0, -60.0, zz]), U, V, 300.0, 80.0, 2.0, 900 + seed, 0, (0.0, 0.0, 0.0)))
    return planes


def kitti_path(n, seed, speed=0.35):
    """Forward motion (speed m/frame) with a gentle sway and yaw (<= 0.3 deg/frame).  Returns Twc [n,4,4]."""
    rng = np.random.Generator(np.random.PCG64(0xFACE + 32452843 * seed))
    ph = rng.uniform(0, 2 * np.pi, 3)
    t = np.arange(n)
    x = 0.6 * np.sin(2 * np.pi * t / 260 + ph[0])
    y = 0.05 * np.sin(2 * np.pi * t / 90 + ph[1])
    zc = speed * t
    yaw = np.deg2rad(4.0) * np.sin(2 * np.pi * t / 210 + ph[2])
    T = np.tile(np.eye(4), (n, 1, 1))
    for i in range(n):
        T[i, :3, :3] = _rot_xyz(0.0, yaw[i], 0.0)
        T[i, :3, 3] = (x[i], y[i], zc[i])
    return T


def make_stereo_sequence(seed, n, width=1241, height=376, speed=0.35, baseline=KITTI_BASELINE, first=0, count=None):
    """S3: dict(gray = left [n,H,W] u8, right [n,H,W] u8, Twc [n,4,4] relative to the first left camera).  first / count: only frames
    [first, first + count) of the n-frame stream (see make_rgbd_sequence)."""
    planes = make_kitti_scene(seed, length=max(120.0, speed * n + 90.0))
    Twc = kitti_path(n, seed, speed)
    m = n - first if count is None else count
    left = np.empty((m, height, width), np.uint8)
    right = np.empty((m, height, width), np.uint8)
    for j in range(m):
        i = first + j
        C = Twc[i, :3, 3]
        near = [p for p in planes if -15.0 < (p.P0[2] - C[2]) < 140.0 or p.density < 5.0]
        left[j] = render(near, Twc[i], KITTI_K, width, height)[0]
        Tr = Twc[i].copy()
        Tr[:3, 3] = C + Twc[i, :3, 0] * baseline          # right camera: +baseline along the camera x axis
        right[j] = render(near, Tr, KITTI_K, width, height)[0]
    T0inv = np.linalg.inv(Twc[0])
    return dict(gray=left, right=right, Twc=np.array([T0inv @ T for T in Twc[first:first + m]]))
