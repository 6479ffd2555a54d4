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
