"""Host-stage timing of the driver WITHOUT a GPU: one S1 sequence through oslam_slam over the CPU oracle's operator table (the driver's host code is the same
under either table), printing the per-stage core-seconds of the host stages and the final statistics.  An A/B tool for changes to the driver's host loops
(the statistics must not change).  Lives under tests/ because it loads the oracle; not collected by pytest.
usage: python tests/host_stage_cpu.py [n=300] [sync|deferred]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
schedule = sys.argv[2] if len(sys.argv) > 2 else "deferred"
CH = 25


def piece(first):
    from object_slam_amd import scene
    return scene.make_rgbd_sequence(0, n, speed=1.0, first=first, count=min(CH, n - first))


def main():
    with mp.get_context("fork").Pool(8) as pool:
        ps = pool.map(piece, list(range(0, n, CH)), chunksize=1)
    q = {k: (np.concatenate([p[k] for p in ps]) if k in ("gray", "depth", "masks", "Twc") else ps[0][k]) for k in ps[0]}
    from object_slam_amd import slam
    from slam_common import H, W, oracle_ops
    cfg = slam.make_config(W, H, 1, local_mapping=slam.LM_DEFERRED if schedule == "deferred" else slam.LM_SYNC)
    sy = slam.System(cfg, oracle_ops(cfg))
    t0 = time.time()
    for t in range(n):
        sy.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=[dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])])
    sy.finish()
    cpu = sy.stage_seconds(cpu=True)
    host = {k: round(v, 4) for k, v in cpu.items() if k.startswith(("h", "mp_"))}
    print(json.dumps(dict(frames=n, wall_s=round(time.time() - t0, 1), host_core_s=host, stats=sy.stats(0), windows=sy.lba_window_stats(0))))


if __name__ == "__main__":
    main()
