"""Seeded synthetic inputs (S1/S2/S3 shapes).

Integer-only rendering so every host produces identical bytes: a large textured canvas
(multi-octave integer value noise + random axis-aligned rectangles, corner rich) seen by a
camera translating parallel to it, i.e. frame t is an integer-offset crop.  The 3-D model
is the fronto-parallel plane z = Z0 in the world frame; the camera pose for frame t is a
pure translation (tx, ty, 0) with u-shift = fx*tx/Z0.
"""
import numpy as np

SEED = 0x0B5E55ED


def make_canvas(width, height, seed=SEED, n_rects=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    canvas = np.zeros((height, width), np.int32)
    # three octaves of integer value noise (nearest-neighbour upsampled blocks)
    for cell, amp in ((64, 48), (16, 24), (4, 12)):
        gh, gw = (height + cell - 1) // cell, (width + cell - 1) // cell
        g = rng.integers(0, amp, size=(gh, gw), dtype=np.int32)
        canvas += np.kron(g, np.ones((cell, cell), np.int32))[:height, :width]
    canvas += 60
    if n_rects is None:
        n_rects = (width * height) // 500
    xs = rng.integers(0, width, n_rects)
    ys = rng.integers(0, height, n_rects)
    ws = rng.integers(3, 60, n_rects)
    hs = rng.integers(3, 60, n_rects)
    gs = rng.integers(0, 256, n_rects)
    for x, y, w, h, g in zip(xs, ys, ws, hs, gs):
        canvas[y:y + h, x:x + w] = g
    # fine grain so flat regions are not exactly constant
    canvas += rng.integers(-3, 4, size=canvas.shape, dtype=np.int32)
    return np.clip(canvas, 0, 255).astype(np.uint8)


def stream_offsets(n_frames, max_dx, max_dy, seed=SEED):
    """Smooth integer crop offsets (a slow Lissajous path inside [0,max_dx]x[0,max_dy])."""
    t = np.arange(n_frames)
    ph = (seed % 997) / 997.0
    ox = np.rint((0.5 + 0.5 * np.sin(2 * np.pi * (t / 240.0 + ph))) * max_dx).astype(np.int64)
    oy = np.rint((0.5 + 0.5 * np.sin(2 * np.pi * (t / 170.0 + 2 * ph))) * max_dy).astype(np.int64)
    return ox, oy


def make_stream(n_frames, width=640, height=480, seed=SEED, margin=96):
    """Returns (frames uint8 [n,h,w], offsets int64 [n,2])."""
    canvas = make_canvas(width + margin, height + margin, seed)
    ox, oy = stream_offsets(n_frames, margin, margin, seed)
    frames = np.empty((n_frames, height, width), np.uint8)
    for i in range(n_frames):
        frames[i] = canvas[oy[i]:oy[i] + height, ox[i]:ox[i] + width]
    return frames, np.stack([ox, oy], 1)


def make_occluded_stream(n_frames, width=640, height=480, seed=SEED, margin=60, period=30, every=2, frac=55):
    """A stream that makes the local mapper cull keyframes: the camera swings back and forth over `margin` pixels (so the map
    is soon complete) while every `every`-th frame has `frac` % of its width, alternately at the right and the left edge,
    covered by foreign texture.  The covered frames track too few of their reference keyframe's points, a keyframe is inserted
    in known territory, and older keyframes become redundant (reference src/LocalMapping.cc:KeyFrameCulling).
    Returns (frames uint8 [n,h,w], offsets int64 [n,2])."""
    canvas = make_canvas(width + margin, height + margin, seed)
    t = np.arange(n_frames)
    ox = np.rint((0.5 - 0.5 * np.cos(2 * np.pi * t / period)) * margin).astype(np.int64)
    oy = np.rint((0.5 - 0.5 * np.cos(2 * np.pi * t / period)) * margin * 0.3).astype(np.int64)
    frames = np.empty((n_frames, height, width), np.uint8)
    ow = width * frac // 100
    for i in range(n_frames):
        frames[i] = canvas[oy[i]:oy[i] + height, ox[i]:ox[i] + width]
        if i % every == every - 1:
            x0 = 0 if (i // every) % 2 else width - ow
            frames[i][:, x0:x0 + ow] = make_canvas(ow, height, seed + 1000 + i)
    return frames, np.stack([ox, oy], 1)


def make_stereo_stream(n_frames, width=1241, height=376, seed=SEED, margin=96, disparity=32):
    """Rectified stereo pair stream of the same fronto-parallel plane (SURVEY.md §8(d) S3 shape): the right camera sees every
    point `disparity` pixels further left (uR = uL - disparity), i.e. its crop starts `disparity` columns to the right; the plane
    depth is Z0 = bf / disparity.  Returns (left uint8 [n,h,w], right uint8 [n,h,w], offsets int64 [n,2])."""
    canvas = make_canvas(width + margin + disparity, height + margin, seed)
    ox, oy = stream_offsets(n_frames, margin, margin, seed)
    left = np.empty((n_frames, height, width), np.uint8)
    right = np.empty((n_frames, height, width), np.uint8)
    for i in range(n_frames):
        left[i] = canvas[oy[i]:oy[i] + height, ox[i]:ox[i] + width]
        right[i] = canvas[oy[i]:oy[i] + height, ox[i] + disparity:ox[i] + disparity + width]
    return left, right, np.stack([ox, oy], 1)


# ---------------------------------------------------------------------------------------------
# Geometric problems for the optimisers: no images needed.
# ---------------------------------------------------------------------------------------------
KITTI_K = (718.856, 718.856, 607.1928, 185.2157, 386.1448)   # reference Examples/Stereo/KITTI00-02.yaml:8-20
TUM_K = (520.908620, 521.007327, 325.141442, 249.701764, 40.0)   # reference Examples/RGB-D/TUM2.yaml


def _rot(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def make_T(rv, t):
    T = np.eye(4)
    T[:3, :3] = _rot(np.asarray(rv, float))
    T[:3, 3] = t
    return T


def level_sigma2(nlevels=8, sf=1.2):
    s = np.float32(1.0)
    out = [np.float32(1.0)]
    for _ in range(1, nlevels):
        s = np.float32(s * np.float32(sf))
        out.append(np.float32(s * s))
    return np.array(out, np.float32)


def project(T, X, K):
    fx, fy, cx, cy, bf = K
    Xc = X @ T[:3, :3].T + T[:3, 3]
    z = Xc[:, 2]
    u = fx * Xc[:, 0] / z + cx
    v = fy * Xc[:, 1] / z + cy
    return u, v, u - bf / z, z


def make_pose_problem(seed, N=1000, K=TUM_K, width=640, height=480, outlier_frac=0.1, stereo_frac=0.7,
                      noise=1.0, pose_err=(0.02, 0.05), mp_frac=0.8):
    """One PoseOptimization input: N keypoints, a fraction with map points."""
    rng = np.random.default_rng(seed)
    T_gt = make_T(rng.normal(0, 0.1, 3), rng.normal(0, 0.3, 3))
    fx, fy, cx, cy, bf = K
    u = rng.uniform(10, width - 10, N)
    v = rng.uniform(10, height - 10, N)
    z = rng.uniform(0.6, 8.0, N)
    Xc = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
    Xw = (Xc - T_gt[:3, 3]) @ T_gt[:3, :3]          # R^T (Xc - t)
    octave = rng.integers(0, 8, N)
    sig2 = level_sigma2()[octave]
    obs = np.stack([u, v, u - bf / z], 1)
    obs[:, :2] += rng.normal(0, noise, (N, 2)) * np.sqrt(sig2)[:, None]
    obs[:, 2] += rng.normal(0, noise, N) * np.sqrt(sig2)
    mono = rng.random(N) > stereo_frac
    obs[mono, 2] = -1.0
    out = rng.random(N) < outlier_frac
    obs[out, 0] += rng.choice([-1, 1], out.sum()) * rng.uniform(15, 60, out.sum())
    has_mp = (rng.random(N) < mp_frac).astype(np.uint8)
    T0 = make_T(rng.normal(0, pose_err[0], 3), rng.normal(0, pose_err[1], 3)) @ T_gt
    return dict(Tcw=T0.astype(np.float32), T_gt=T_gt, Xw=Xw.astype(np.float32), obs=obs.astype(np.float32),
                invSigma2=(np.float32(1.0) / sig2).astype(np.float32), has_mp=has_mp, K=np.array(K, np.float32),
                is_outlier=out)


def make_lba_problem(seed, K_local=20, K_fixed=20, P=4000, track=6, K=KITTI_K, width=1241, height=376,
                     outlier_frac=0.05, noise=1.0, pose_err=(0.005, 0.02), point_err=0.05, stereo_frac=0.85):
    """S5-shaped local BA: keyframes along a forward path, each point seen by ~track keyframes."""
    rng = np.random.default_rng(seed)
    nKF = K_local + K_fixed
    Ts = []
    for k in range(nKF):
        pos = np.array([0.05 * rng.normal(), 0.02 * rng.normal(), 0.6 * k])
        R = _rot(np.array([0.01 * rng.normal(), 0.03 * np.sin(k / 5.0) + 0.01 * rng.normal(), 0.005 * rng.normal()]))
        Twc = np.eye(4)
        Twc[:3, :3] = R
        Twc[:3, 3] = pos
        Ts.append(np.linalg.inv(Twc))
    Ts = np.array(Ts)
    fixed = np.zeros(nKF, np.uint8)
    # the oldest keyframes are the fixed cameras; keyframe id 0 is local-but-fixed when no fixed set
    fixed[:K_fixed] = 1
    if K_fixed == 0:
        fixed[0] = 2
    fx, fy, cx, cy, bf = K
    pts, ekf, ept, eobs, einv = [], [], [], [], []
    sig2 = level_sigma2()
    pid = 0
    while pid < P:
        k0 = int(rng.integers(0, nKF))
        Twc = np.linalg.inv(Ts[k0])
        z = rng.uniform(4, 40)
        u, v = rng.uniform(20, width - 20), rng.uniform(20, height - 20)
        Xw = Twc[:3, :3] @ np.array([(u - cx) * z / fx, (v - cy) * z / fy, z]) + Twc[:3, 3]
        seen = []
        for k in range(max(0, k0 - track), min(nKF, k0 + track + 1)):
            uu, vv, ur, zz = project(Ts[k], Xw[None], K)
            if zz[0] > 0.5 and 0 <= uu[0] < width and 0 <= vv[0] < height and rng.random() < 0.5:
                seen.append((k, uu[0], vv[0], ur[0]))
        if len(seen) < 2 or not any(fixed[k] == 0 for k, *_ in seen):
            continue
        pts.append(Xw)
        for k, uu, vv, ur in seen:
            o = int(rng.integers(0, 8))
            s = np.sqrt(sig2[o]) * noise
            ob = [uu + rng.normal(0, s), vv + rng.normal(0, s), ur + rng.normal(0, s)]
            if rng.random() > stereo_frac:
                ob[2] = -1.0
            if rng.random() < outlier_frac:
                ob[0] += rng.choice([-1, 1]) * rng.uniform(20, 50)
            ekf.append(k); ept.append(pid); eobs.append(ob); einv.append(np.float32(1.0) / sig2[o])
        pid += 1
    pts = np.array(pts)
    poses0 = []
    for k in range(nKF):
        if fixed[k]:
            poses0.append(Ts[k])
        else:
            poses0.append(make_T(rng.normal(0, pose_err[0], 3), rng.normal(0, pose_err[1], 3)) @ Ts[k])
    return dict(poses=np.array(poses0, np.float32), poses_gt=Ts, fixed=fixed,
                points=(pts + rng.normal(0, point_err, pts.shape)).astype(np.float32), points_gt=pts,
                edge_kf=np.array(ekf, np.int32), edge_pt=np.array(ept, np.int32), edge_obs=np.array(eobs, np.float32),
                edge_invSigma2=np.array(einv, np.float32), K=np.array(K, np.float32))


def make_semantic_problem(seed, N=1000, n_obj=3, K=TUM_K, width=640, height=480, **kw):
    """PoseOptimization2 input: a pose problem plus object masks (boxes with ragged edges), object map
    points (projecting in / near their mask) and the M_joint set (keypoints with a map point of an
    object that lie just outside its mask)."""
    p = make_pose_problem(seed, N=N, K=K, width=width, height=height, **kw)
    rng = np.random.default_rng(seed + 77)
    fx, fy, cx, cy, bf = K
    T = p["T_gt"]
    masks = np.zeros((n_obj, height, width), np.uint8)
    boxes = []
    for o in range(n_obj):
        w, h = rng.integers(60, 180), rng.integers(60, 160)
        x0, y0 = rng.integers(20, width - w - 20), rng.integers(20, height - h - 20)
        masks[o, y0:y0 + h, x0:x0 + w] = 255
        # ragged border so nearest pixels are not trivially axis aligned
        for _ in range(40):
            rx, ry = rng.integers(x0, x0 + w), rng.integers(y0, y0 + h)
            masks[o, max(ry - 3, 0):ry + 3, max(rx - 3, 0):rx + 3] = rng.choice([0, 255])
        boxes.append((x0, y0, w, h))
    objmp_Xw, objmp_obj = [], []
    for o, (x0, y0, w, h) in enumerate(boxes):
        m = int(rng.integers(30, 120))
        u = rng.uniform(x0 - 12, x0 + w + 12, m)
        v = rng.uniform(y0 - 12, y0 + h + 12, m)
        z = rng.uniform(1.0, 5.0, m)
        Xc = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
        Xw = (Xc - T[:3, 3]) @ T[:3, :3]
        objmp_Xw.append(Xw)
        objmp_obj += [o] * m
    kp_uv = p["obs"][:, :2].copy()
    joint_kp, joint_obj = [], []
    for o, (x0, y0, w, h) in enumerate(boxes):
        near = np.where((p["has_mp"] > 0) & (kp_uv[:, 0] > x0 - 15) & (kp_uv[:, 0] < x0 + w + 15) & (kp_uv[:, 1] > y0 - 15) &
                        (kp_uv[:, 1] < y0 + h + 15))[0]
        sel = near[rng.random(len(near)) < 0.5]
        joint_kp += list(sel)
        joint_obj += [o] * len(sel)
    p.update(masks=masks, objmp_Xw=np.concatenate(objmp_Xw).astype(np.float32), objmp_obj=np.array(objmp_obj, np.int32),
             joint_kp=np.array(joint_kp, np.int32), joint_obj=np.array(joint_obj, np.int32), kp_uv=kp_uv.astype(np.float32),
             bounds=np.array([0, 0, width, height], np.float32), invSigma2_0=np.float32(1.0))
    return p
