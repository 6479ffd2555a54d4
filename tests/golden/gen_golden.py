#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle on small seeded inputs.

The reference has no golden vectors and cannot be built or imported here, so
these fixtures pin the ORACLE against regressions (and travel to the GPU box as data); they are not
outputs of the reference.  Run from the repo root: python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from object_slam_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    frames, offs = synth.make_stream(2, 320, 240, seed=1234, margin=32)
    e = O.OrbExtractor(300, 1.2, 5, 20, 7)
    k0, d0 = e.extract(frames[0])
    k1, d1 = e.extract(frames[1])
    sf = e.tables()["scale"]
    fx = fy = 260.0
    cx, cy, bf, Z0 = 160.0, 120.0, 20.0, 2.0
    cam = np.array([fx, fy, cx, cy, bf, bf / fx], np.float32)
    Xw = np.stack([(k0["x"] - cx) * Z0 / fx, (k0["y"] - cy) * Z0 / fy, np.full(len(k0), Z0)], 1).astype(np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    du, dv = (offs[1] - offs[0]).astype(np.float64)
    Tcw[0, 3], Tcw[1, 3] = -du * Z0 / fx, -dv * Z0 / fy
    Tlw = np.eye(4, dtype=np.float32)
    has = np.full(len(k0), 3, np.uint8)
    has[::7] = 1
    has[::11] = 0
    bounds = np.array([0, 0, 320, 240], np.float32)
    q = O.project_last_frame(Xw, has, k0, d0, Tcw, Tlw, cam, bounds, sf, 15.0, False)
    uR = (k1["x"] - bf / Z0).astype(np.float32)
    nm, qm, qd, km = O.search_by_projection(k1, uR, d1, None, bounds, q, 0.9, False, True)
    nm2, qm2, qd2, km2 = O.search_by_projection(k1, uR, d1, None, bounds, q, 0.8, True, False)
    np.savez_compressed(os.path.join(OUT, "frontend_320x240.npz"), frame0=frames[0], frame1=frames[1], offs=offs,
                        k0=k0, d0=d0, k1=k1, d1=d1, scale=sf, cam=cam, Xw=Xw, Tcw=Tcw, Tlw=Tlw, has=has, bounds=bounds,
                        queries=q, uR=uR, nm=nm, qm=qm, qd=qd, km=km, nm2=nm2, qm2=qm2, qd2=qd2, km2=km2)
    print("frontend_320x240.npz: %d / %d keypoints, %d / %d matches" % (len(k0), len(k1), nm, nm2))


if __name__ == "__main__":
    main()
