"""Latency of ONE sequence through the driver (what a drop-in Tracking / LocalMapping caller sees): S1 stream with masks, host images.
usage: python tools/single_seq.py [n=150] [masks=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiprocessing as mp
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
with_masks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CH = 10


def piece(first):
    from object_slam_amd import scene
    return scene.make_rgbd_sequence(0, n, speed=1.0, first=first, count=min(CH, n - first))


if __name__ == "__main__":
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:
        parts = pool.map(piece, list(range(0, n, CH)))
    q = {k: (np.concatenate([p[k] for p in parts]) if k in ("gray", "depth", "masks", "Twc") else parts[0][k]) for k in parts[0]}
    from object_slam_amd import slam
    for threads in (1, 4):
        sysm = slam.System(slam.make_config(640, 480, 1, host_threads=threads))
        per = []
        for t in range(n):
            objs = [dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])] if with_masks else None
            t0 = time.perf_counter()
            T, st = sysm.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
            per.append(time.perf_counter() - t0)
        per = np.array(per[5:]) * 1e3
        stt = sysm.stats(0)
        print("host_threads %d: mean %.2f ms, median %.2f ms, p95 %.2f ms per frame (%.0f frames/s); keyframes %d, local BAs %d, lost %d"
              % (threads, per.mean(), np.median(per), np.percentile(per, 95), 1e3 / per.mean(), stt["keyframes_created"], stt["local_bas"], stt["lost_frames"]))
        print("  stage ms per frame:", {k: round(v / n * 1e3, 3) for k, v in sysm.stage_seconds().items() if v > 0})
        sysm.close()
