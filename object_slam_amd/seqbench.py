"""Batch-of-sequences runner (BASELINE.json configs[4]): every rank owns a shard of independent sequences
(sequence i -> rank i mod N, `parallel.shard_sequences`) and advances them through the tracking + local-mapping driver
(include/oslam_slam.h) with no data-path collective; torch.distributed (RCCL on the GPU box, gloo in CPU tests) is used for the
barriers around the timed region, the max-over-ranks wall time and ONE all-gather of a fixed-size stats record per rank.

bench.py and tests/test_parallel_gloo.py call the same entry point, `run_rank`; the operator table comes from `make_system`
(the product binds the HIP table; the gloo test passes a factory that binds the CPU oracle's table — nothing here imports oracle/).
"""
import threading
import time

import numpy as np

from . import slam
from .e2e import horn_align_ate
from .parallel import aggregate_stats, gather_records, shard_sequences

RECORD_FIELDS = ("rank", "sequences", "frames", "elapsed_s", "keyframes", "local_bas", "lost_frames", "ate_rmse_m", "map_violations", "semantic_edges")


class Workload:
    """Shape of one batch-of-sequences run.  `make_sequence(seq_id, n_frames)` returns a dict with `gray` [n,H,W] uint8, `Twc` [n,4,4]
    ground truth and, for RGB-D, `depth` [n,H,W] float32 (metres) (+ optional `masks` [n,nObj,H,W] uint8 and `track_ids` [nObj]); for stereo, `right`."""

    def __init__(self, name, width, height, sensor, cam, nFeatures, fps, make_sequence, n_base=8, stagger=8):
        self.name, self.width, self.height, self.sensor, self.cam = name, width, height, sensor, cam
        self.nFeatures, self.fps, self.make_sequence, self.n_base, self.stagger = nFeatures, fps, make_sequence, n_base, stagger


class _Inputs:
    """Frames of the rank's base sequences, on the host or resident in HBM.  Sequence g (global id) replays base sequence
    g mod n_base from frame offset (g div n_base) mod (stagger + 1): sequences of one handle see different frames at the same step,
    so keyframe insertions and local BAs are spread over the steps instead of hitting every sequence at once."""

    def __init__(self, wl, seq_ids, n_frames, on_device, device=None, base_seed=0, sequences=None):
        self.wl, self.on_device = wl, on_device
        self.length = n_frames + wl.stagger
        # the i-th sequence of this rank: base sequence i mod n_base (seeded per rank), frame offset (i div n_base) mod (stagger + 1)
        self.base = {g: i % wl.n_base for i, g in enumerate(seq_ids)}
        self.off = {g: (i // wl.n_base) % (wl.stagger + 1) for i, g in enumerate(seq_ids)}
        self.base_ids = sorted(set(self.base.values()))
        self.seqs = sequences if sequences is not None else {b: wl.make_sequence(base_seed + b, self.length) for b in self.base_ids}
        self.pitch = (wl.width + 63) // 64 * 64 if on_device else wl.width
        if on_device:
            import torch
            self.dev = {}
            for b, q in self.seqs.items():
                d = {}
                for key in ("gray", "right"):
                    if key in q:
                        t = torch.zeros((self.length, wl.height, self.pitch), dtype=torch.uint8, device=device)
                        t[:, :, :wl.width] = torch.from_numpy(q[key]).to(device)
                        d[key] = t
                if "depth" in q:
                    d["depth"] = torch.from_numpy(np.ascontiguousarray(q["depth"])).to(device)
                if q.get("masks") is not None:
                    d["masks"] = torch.from_numpy(np.ascontiguousarray(q["masks"])).to(device)
                self.dev[b] = d
            torch.cuda.synchronize()

    def frame(self, g, t, key):
        b, tt = self.base[g], t + self.off[g]
        if self.on_device:
            x = self.dev[b][key][tt]
            return x.data_ptr()
        return self.seqs[b][key][tt]

    @property
    def has_masks(self):
        return any(q.get("masks") is not None for q in self.seqs.values())

    def objects(self, g, t):
        """The frame's detections for slam.System.TrackRGBD: the three instance masks of the scene with their ground-truth identities."""
        b, tt = self.base[g], t + self.off[g]
        q = self.seqs[b]
        if q.get("masks") is None:
            return None
        n = q["masks"].shape[1]
        if self.on_device:
            m = self.dev[b]["masks"]
            masks = [m[tt, o].data_ptr() for o in range(n)]
        else:
            masks = [q["masks"][tt, o] for o in range(n)]
        return dict(masks=masks, track_ids=q["track_ids"], labels=q.get("labels", [0] * n))

    def gt(self, g, n):
        b, o = self.base[g], self.off[g]
        T = self.seqs[b]["Twc"][o:o + n]
        T0inv = np.linalg.inv(T[0])
        return np.array([T0inv @ x for x in T])


def _prepare(system, wl, inp, seq_ids, n_frames):
    """The call arguments of every step of one handle, marshalled once (the frame loop then only crosses the C boundary)."""
    S = len(seq_ids)
    out = []
    for t in range(n_frames):
        stamps = [t / wl.fps] * S
        if wl.sensor == slam.STEREO:
            out.append(system.prepare_stereo([inp.frame(g, t, "gray") for g in seq_ids], [inp.frame(g, t, "right") for g in seq_ids], stamps,
                                             on_device=inp.on_device, stride=inp.pitch))
        else:
            objs = [inp.objects(g, t) for g in seq_ids] if inp.has_masks else None
            out.append(system.prepare_rgbd([inp.frame(g, t, "gray") for g in seq_ids], [inp.frame(g, t, "depth") for g in seq_ids], stamps, objects=objs,
                                           on_device=inp.on_device, gray_stride=inp.pitch, depth_pitch=wl.width, mask_stride=wl.width))
    return out


def _drive(system, wl, calls, t0, t1, poses_out=None):
    for t in range(t0, t1):
        T, _ = system.track_stereo_prepared(calls[t]) if wl.sensor == slam.STEREO else system.track_prepared(calls[t])
        if poses_out is not None:
            poses_out.append(T.copy())


def base_sequences(wl, rank, seqs_per_rank, n_frames, workers=1):
    """The rank's base sequences {b: dict}, rendered by `workers` processes (call it before the process touches the GPU: fork)."""
    nb = min(wl.n_base, seqs_per_rank)
    jobs = [(wl.n_base * rank + b, n_frames + wl.stagger) for b in range(nb)]
    if workers > 1 and nb > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, nb)) as pool:
            out = pool.starmap(wl.make_sequence, jobs)
    else:
        out = [wl.make_sequence(*j) for j in jobs]
    return dict(enumerate(out))


def run_rank(wl, make_system, rank, world, seqs_per_rank, handles, steps, warmup, on_device, device=None, host_threads=0, collect_poses=False,
             sequences=None, after_warmup=None, coll_on_device=True):
    """Runs this rank's shard: `seqs_per_rank` sequences in `handles` driver handles (each advanced by its own host thread), `warmup`
    untimed lockstep steps, then exactly `steps` timed steps bracketed by a barrier + device synchronisation on both sides.
    Returns (summary dict on every rank, per-rank records [world, len(RECORD_FIELDS)], systems, extra)."""
    import torch
    import torch.distributed as dist
    multi = world > 1 and dist.is_available() and dist.is_initialized()
    total = seqs_per_rank * world
    mine = shard_sequences(total, world, rank)                 # global sequence ids of this rank
    assert len(mine) == seqs_per_rank and seqs_per_rank % handles == 0
    per = seqs_per_rank // handles
    groups = [mine[h * per:(h + 1) * per] for h in range(handles)]
    n_frames = warmup + steps
    inp = _Inputs(wl, mine, n_frames, on_device, device, base_seed=wl.n_base * rank, sequences=sequences)
    systems = []
    for h in range(handles):
        cfg = slam.make_config(wl.width, wl.height, per, cam=wl.cam, nFeatures=wl.nFeatures, sensor=wl.sensor,
                               device=(device.index if hasattr(device, "index") and device.index is not None else 0) if on_device else 0, host_threads=host_threads)
        systems.append(make_system(cfg))
    poses = [[] for _ in range(handles)] if collect_poses else [None] * handles
    calls = [_prepare(systems[h], wl, inp, groups[h], n_frames) for h in range(handles)]

    def sync():
        if on_device:
            torch.cuda.synchronize()

    def phase(t0, t1):
        ths = [threading.Thread(target=_drive, args=(systems[h], wl, calls[h], t0, t1, poses[h])) for h in range(handles)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    phase(0, warmup)
    sync()
    if after_warmup is not None:
        after_warmup(systems)
    if multi:
        dist.barrier()
    sync()
    tstart = time.perf_counter()
    phase(warmup, n_frames)
    sync()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - tstart
    frames = seqs_per_rank * steps
    total_frames, max_elapsed = aggregate_stats(elapsed, frames, device=device if (on_device and coll_on_device) else None)

    # per-rank record (fixed size, all-gathered once): counters summed over the rank's sequences, ATE of its first sequence
    kf = lba = lost = viol = sem = 0
    for h in range(handles):
        for s in range(per):
            st = systems[h].stats(s)
            kf += st["keyframes_created"]; lba += st["local_bas"]; lost += st["lost_frames"]; viol += st["map_violations"]
            sem += st.get("semantic_edges", 0)
    _, Twc = systems[0].trajectory(0)
    gt = inp.gt(groups[0][0], n_frames)
    ate = horn_align_ate(Twc[:, :, 3], gt[:len(Twc), :3, 3]) if len(Twc) >= 3 else float("nan")
    rec = [rank, seqs_per_rank, frames, elapsed, kf, lba, lost, ate, viol, sem]
    records = gather_records(rec, device=device if (on_device and coll_on_device) else None).numpy()
    summary = {"frames_per_s": total_frames / max_elapsed, "total_frames": total_frames, "elapsed_s": max_elapsed,
               "ms_per_step": max_elapsed / steps * 1e3, "n_ranks": int(records.shape[0]),
               "keyframes": int(records[:, 4].sum()), "local_bas": int(records[:, 5].sum()), "lost_frames": int(records[:, 6].sum()),
               "ate_rmse_m": float(np.nanmean(records[:, 7])), "map_violations": int(records[:, 8].sum()), "semantic_edges": int(records[:, 9].sum())}
    return summary, records, systems, {"inputs": inp, "groups": groups, "poses": poses}


# ---- the two stream shapes of SURVEY.md §8(d) ----
def _make_rgbd(seed, n, speed=2.0, with_masks=True):
    from . import scene
    return scene.make_rgbd_sequence(seed, n, speed=speed, with_masks=with_masks)


def _make_stereo(seed, n, speed=0.35):
    from . import scene
    return scene.make_stereo_sequence(seed, n, speed=speed)


def rgbd_workload(speed=2.0, with_masks=True, n_base=8, stagger=8):
    """S1: TUM-shaped RGB-D room with three box objects (640x480, 1000 features, TUM2.yaml calibration)."""
    import functools
    return Workload("S1 TUM-shaped RGB-D", 640, 480, slam.RGBD, slam.TUM2, 1000, 30.0, functools.partial(_make_rgbd, speed=speed, with_masks=with_masks), n_base, stagger)


def stereo_workload(speed=0.35, n_base=4, stagger=4):
    """S3/S4: KITTI-shaped rectified stereo street (1241x376, 2000 features, KITTI00-02.yaml calibration)."""
    import functools
    return Workload("S3 KITTI-shaped stereo", 1241, 376, slam.STEREO, slam.KITTI00, 2000, 10.0, functools.partial(_make_stereo, speed=speed), n_base, stagger)
