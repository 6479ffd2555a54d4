"""S1 soak: the 1000-frame TUM-shaped RGB-D stream with three instance masks through the oslam_slam driver on the
HIP operator table — S sequences in lockstep — and, for sequence 0, through the SAME driver over the CPU oracle's operator table; reports frames/s,
ATE (Horn alignment, evaluate_ate.py semantics), lost frames, map statistics and the HIP-vs-oracle agreement.
Lives under tests/ because its comparison leg loads the oracle (test infrastructure); not collected by pytest.
The tables agree on every statistic for as long as no thresholded decision falls inside the optimisers' tolerance: `first_stat_difference`
reports the first frame at which a counter of sequence 0 differs, what differed, and how far apart the poses were up to there.
usage: python tests/soak_s1.py [S=2] [n=1000] [oracle=1] [sync|deferred]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import numpy as np

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
with_oracle = int(sys.argv[3]) if len(sys.argv) > 3 else 1
schedule = sys.argv[4] if len(sys.argv) > 4 else "sync"
CH = 25


def piece(a):
    from object_slam_amd import scene
    seed, first = a
    return scene.make_rgbd_sequence(seed, n, speed=1.0, first=first, count=min(CH, n - first))


def main():
    t0 = time.time()
    jobs = [(s, f) for s in range(S) for f in range(0, n, CH)]
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:   # before anything touches the GPU
        parts = pool.map(piece, jobs, chunksize=1)
    seqs = []
    for s in range(S):
        ps = [p for (ss, _), p in zip(jobs, parts) if ss == s]
        seqs.append({k: (np.concatenate([p[k] for p in ps]) if k in ("gray", "depth", "masks", "Twc") else ps[0][k]) for k in ps[0]})
    del parts
    print("rendered", S, "x", n, "frames in", round(time.time() - t0, 1), "s", flush=True)
    from object_slam_amd import slam
    from object_slam_amd.io import horn_align_ate
    from slam_common import H, W, oracle_ops

    LM = slam.LM_DEFERRED if schedule == "deferred" else slam.LM_SYNC

    def drive(system, ss, label, snaps=None):
        poses, t1 = [], time.time()
        for t in range(n):
            objs = [dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"]) for q in ss]
            T, st = system.TrackRGBD([q["gray"][t] for q in ss], [q["depth"][t] for q in ss], [t / 30.0] * len(ss), objects=objs)
            poses.append(T.copy())
            if snaps is not None:
                snaps.append(system.stats(0))
            if (t + 1) % 100 == 0:
                print(label, "frame", t + 1, "states", st.tolist(), round(time.time() - t1, 1), "s", flush=True)
        return np.array(poses), time.time() - t1

    hip = slam.System(slam.make_config(W, H, S, host_threads=8, local_mapping=LM))
    sh = []
    ph, dt = drive(hip, seqs, "hip", sh)
    out = {"sequences": S, "frames": n, "schedule": schedule, "hip_frames_per_s": round(S * n / dt, 1), "per_sequence": []}
    for s in range(S):
        stamps, Twc = hip.trajectory(s)
        a = horn_align_ate(Twc[:, :, 3], seqs[s]["Twc"][:len(stamps), :3, 3])
        out["per_sequence"].append(dict(ate_rmse_m=round(float(a), 5), **hip.stats(s)))
    if with_oracle:
        cfg = slam.make_config(W, H, 1, local_mapping=LM)
        ora = slam.System(cfg, oracle_ops(cfg))
        so = []
        po, dto = drive(ora, seqs[:1], "oracle", so)
        stamps, Twc = ora.trajectory(0)
        out["oracle_seq0"] = dict(frames_per_s=round(n / dto, 1), ate_rmse_m=round(float(horn_align_ate(Twc[:, :, 3], seqs[0]["Twc"][:len(stamps), :3, 3])), 5),
                                  **ora.stats(0))
        out["hip_vs_oracle_seq0"] = dict(stats_equal=hip.stats(0) == ora.stats(0), max_abs_pose_diff=float(np.abs(ph[:, 0] - po[:, 0]).max()))
        first = next((t for t in range(n) if sh[t] != so[t]), None)
        if first is not None:
            out["first_stat_difference"] = dict(frame=first, counters={k: (sh[first][k], so[first][k]) for k in sh[first] if sh[first][k] != so[first][k]},
                                                max_abs_pose_diff_before=float(np.abs(ph[:first + 1, 0] - po[:first + 1, 0]).max()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
