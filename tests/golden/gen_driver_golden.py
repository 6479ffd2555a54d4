#!/usr/bin/env python3
"""Generates tests/golden/driver_*.npz: per-frame states, poses and map statistics of oracle/slam_driver_oracle.py — the object-style PYTHON restatement of
the reference's Tracking / LocalMapping flow (src/Tracking.cc:310-587, src/LocalMapping.cc:48-113) over the CPU oracle's operators — on seeded streams.

Purpose: on the GPU box the product's C++ driver over the HIP operators is otherwise only compared with the same C++ driver over the oracle's
operators; these fixtures put an independent restatement of the DRIVER on the other side of the comparison (tests/test_slam_driver_gpu.py).  The reference
itself cannot be built or imported here: the fixtures are outputs of the oracle, not of the reference.
Run from the repo root: python tests/golden/gen_driver_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from object_slam_amd import slam  # noqa: E402
from oracle import slam_driver_oracle as R  # noqa: E402
from slam_common import H, W, make_scene_streams, make_streams  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STAT_KEYS = ["frames", "keyframes_created", "keyframes_in_map", "points_created", "points_in_map", "local_bas", "tracked_motion_model", "tracked_reference_kf", "lost_frames",
             "points_fused", "points_triangulated", "keyframes_culled", "points_culled", "last_inliers", "lba_edges"]


def cfg_dict(cfg):
    return dict(width=cfg.width, height=cfg.height, fx=cfg.fx, fy=cfg.fy, cx=cfg.cx, cy=cfg.cy, bf=cfg.bf, thDepth=cfg.thDepth, fps=cfg.fps,
                nFeatures=cfg.nFeatures, scaleFactor=cfg.scaleFactor, nLevels=cfg.nLevels, iniThFAST=cfg.iniThFAST, minThFAST=cfg.minThFAST,
                sensor=cfg.sensor, local_mapping=cfg.local_mapping)


def run(name, lm, n, frames, depth_of, objects_of=None, extra_keys=()):
    cfg = slam.make_config(W, H, 1, local_mapping=lm)
    ref = R.Slam(cfg_dict(cfg))
    keys = STAT_KEYS + list(extra_keys)
    states, poses, stats = [], [], []
    for t in range(n):
        kw = {"objects": objects_of(t)} if objects_of else {}
        T, st = ref.Track((frames[t], depth_of(t)), t / 30.0, **kw)
        states.append(st); poses.append(np.asarray(T, np.float32).copy())
        s = ref.stats()
        stats.append([int(s[k]) for k in keys])
    if hasattr(ref, "FinishLocalMapping"):
        ref.FinishLocalMapping()
    tr = ref.trajectory()
    np.savez_compressed(os.path.join(OUT, name), states=np.array(states, np.int32), poses=np.array(poses, np.float32), stats=np.array(stats, np.int64),
                        stat_keys=np.array(keys), trajectory=np.stack([x[1] for x in tr]).astype(np.float32), local_mapping=np.int32(lm), n=np.int32(n))
    print(name, "frames", n, "final", dict(zip(keys, stats[-1])))


def main():
    n = 40
    streams = make_streams(1, n)
    depth = np.full((H, W), 2.0, np.float32)
    for tag, lm in (("sync", slam.LM_SYNC), ("deferred", slam.LM_DEFERRED)):
        run("driver_rgbd_%s.npz" % tag, lm, n, streams[0][0], lambda t: depth)
    q = make_scene_streams(1, 24)[0]
    run("driver_semantic_sync.npz", slam.LM_SYNC, 24, q["gray"], lambda t: q["depth"][t],
        objects_of=lambda t: dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"]),
        extra_keys=("semantic_edges", "semantic_frames", "object3ds", "object_points", "object2ds"))


if __name__ == "__main__":
    main()
