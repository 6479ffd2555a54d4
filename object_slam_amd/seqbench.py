"""Batch-of-sequences runner (BASELINE.json configs[4]): every rank owns a shard of independent sequences
(sequence i -> rank i mod N, `parallel.shard_sequences`) and advances them through the tracking + local-mapping driver
(include/oslam_slam.h) with no data-path collective; torch.distributed (RCCL on the GPU box, gloo in CPU tests) is used for the
barriers around the timed region, the max-over-ranks wall time and ONE all-gather of a fixed-size stats record per rank.

bench.py and tests/test_parallel_gloo.py call the same entry point, `run_rank`; the operator table comes from `make_system`
(the product binds the HIP table; the gloo test passes a factory that binds the CPU oracle's table — nothing here imports oracle/).
"""
import os
import threading
import time

import numpy as np

from . import slam
from .io import horn_align_ate
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
                        t[:, :, :wl.width] = torch.from_numpy(q[key][:self.length]).to(device)
                        d[key] = t
                if "depth" in q:
                    d["depth"] = torch.from_numpy(np.ascontiguousarray(q["depth"][:self.length])).to(device)
                if q.get("masks") is not None:
                    d["masks"] = torch.from_numpy(np.ascontiguousarray(q["masks"][:self.length])).to(device)
                self.dev[b] = d
            torch.cuda.synchronize()

    def frame(self, g, t, key):
        b, tt = self.base[g], t + self.off[g]
        if self.on_device:
            x = self.dev[b][key][tt]
            return x.data_ptr()
        return self.seqs[b][key][tt]

    def _base_addr(self, b, key, host=False):
        """(address of frame 0, bytes per frame) of plane `key` of base sequence b."""
        if self.on_device and not host:
            t = self.dev[b][key]
            return t.data_ptr(), t.stride(0) * t.element_size()
        a = self.seqs[b][key]
        assert a.flags["C_CONTIGUOUS"]
        return a.__array_interface__["data"][0], a.strides[0]

    def ptr_table(self, seq_ids, key, n_frames, sub=None, host=False):
        """uint64 [n_frames, S] addresses of plane `key` (mask `sub` of the frame's masks when given) of every sequence's frame at every step."""
        S = len(seq_ids)
        base = np.zeros(S, np.uint64); stride = np.zeros(S, np.uint64); off = np.zeros(S, np.uint64)
        for i, g in enumerate(seq_ids):
            a, st = self._base_addr(self.base[g], key, host)
            if sub is not None:
                q = self.dev[self.base[g]][key] if (self.on_device and not host) else self.seqs[self.base[g]][key]
                a += sub * ((q.stride(1) * q.element_size()) if (self.on_device and not host) else q.strides[1])
            base[i], stride[i], off[i] = a, st, self.off[g]
        t = np.arange(n_frames, dtype=np.uint64)[:, None]
        return base[None, :] + (t + off[None, :]) * stride[None, :]

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
    """The call arguments of every step of one handle, marshalled once and vectorised: the pointer tables of all steps are numpy arrays (a Python loop over
    4096 sequences x 250 steps would take longer than the run), the frame loop then only crosses the C boundary."""
    S = len(seq_ids)
    dev = 1 if inp.on_device else 0
    stamps = np.repeat((np.arange(n_frames, dtype=np.float64) / wl.fps)[:, None], S, 1)
    if wl.sensor == slam.STEREO:
        return system.prepare_stereo_bulk(inp.ptr_table(seq_ids, "gray", n_frames), inp.ptr_table(seq_ids, "right", n_frames), stamps, inp.pitch, on_device=dev)
    masks = tids = labs = None
    if inp.has_masks:
        q0 = inp.seqs[inp.base[seq_ids[0]]]
        nobj = q0["masks"].shape[1]
        masks = np.stack([inp.ptr_table(seq_ids, "masks", n_frames, sub=o) for o in range(nobj)], 2)
        tids = np.array([inp.seqs[inp.base[g]]["track_ids"] for g in seq_ids], np.int32)
        labs = np.array([inp.seqs[inp.base[g]].get("labels", [0] * nobj) for g in seq_ids], np.int32)
    return system.prepare_rgbd_bulk(inp.ptr_table(seq_ids, "gray", n_frames), inp.ptr_table(seq_ids, "depth", n_frames), stamps, inp.pitch, wl.width,
                                    masks=masks, track_ids=tids, labels=labs, mask_stride=wl.width, on_device=dev)


def _drive(system, wl, calls, t0, t1, poses_out=None):
    for t in range(t0, t1):
        T, _ = system.track_stereo_prepared(calls[t]) if wl.sensor == slam.STEREO else system.track_prepared(calls[t])
        if poses_out is not None:
            poses_out.append(T.copy())


def _render_chunk(wl, seed, n, first, count):
    return wl.make_sequence(seed, n, first=first, count=count)


def base_sequences(wl, rank, seqs_per_rank, n_frames, workers=1, chunk=24, have=None):
    """The rank's base sequences {b: dict}, rendered by `workers` processes in chunks of `chunk` frames (the images of a frame do not depend on the other
    frames).  Call it before the process touches the GPU: the workers are forked.  `have` = base sequences already rendered for the same stream seeds and
    length (a run of the same workload with fewer bases on rank 0): taken over instead of rendered again."""
    nb = min(wl.n_base, seqs_per_rank)
    n = n_frames + wl.stagger
    have = {b: q for b, q in (have or {}).items() if b < nb and len(q["gray"]) == n and wl.n_base * rank + b == q.get("stream_seed", -1)}
    todo = [b for b in range(nb) if b not in have]
    if not todo:
        return dict(have)
    jobs = [(wl, wl.n_base * rank + b, n, f, min(chunk, n - f)) for b in todo for f in range(0, n, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.starmap(_render_chunk, jobs)
    else:
        parts = [_render_chunk(*j) for j in jobs]
    out = dict(have)
    per = len(jobs) // len(todo)
    for i, b in enumerate(todo):
        ps = parts[i * per:(i + 1) * per]
        q = dict(ps[0])
        for key in ("gray", "right", "depth", "masks", "Twc"):
            if q.get(key) is not None:
                q[key] = np.ascontiguousarray(np.concatenate([p[key] for p in ps], 0))
        q["stream_seed"] = wl.n_base * rank + b
        out[b] = q
    return out


_SHARED_KEYS = ("gray", "right", "depth", "masks", "Twc")


def shared_base_sequences(wl, local_rank, local_world, seqs_per_rank, n_frames, tag, workers=1, shm_dir="/dev/shm", timeout_s=900.0, log=None):
    """The base streams of ONE node rendered once and mapped by all of its ranks: local rank 0 renders rank 0's base streams
    (`base_sequences(wl, 0, ...)`) straight into files under `shm_dir` (tmpfs: one copy in memory for the node, page-shared by the ranks' read-only mappings),
    the other local ranks wait for its `ready` marker and map the same files.  Every rank then replays the SAME base streams (weak scaling: identical work per
    rank); without sharing, rank r renders its own seeds `n_base * r + b`.  `tag` must be the same on the node's ranks and unique per job (bench.py:
    MASTER_PORT).  The files are unlinked as soon as every local rank has mapped them (the mappings stay valid), so nothing is left behind if a rank dies later;
    a rank that cannot get the segment within `timeout_s` raises TimeoutError (bench.py then renders privately)."""
    import json
    import time
    base = os.path.join(shm_dir, "oslam_bases_%s_%s" % (tag, wl.name.replace(" ", "_")))
    ready, meta_path = base + ".ready", base + ".meta.json"

    def mapped_marker(r):
        return base + ".mapped%d" % r

    if local_rank == 0:
        for stale in [ready, meta_path] + [mapped_marker(r) for r in range(local_world)]:
            if os.path.exists(stale):
                os.unlink(stale)
        seqs = base_sequences(wl, 0, seqs_per_rank, n_frames, workers=workers)
        meta = {}
        out = {}
        for b, q in seqs.items():
            mb, qb = {}, {}
            for k, v in q.items():
                if isinstance(v, np.ndarray):
                    path = "%s.b%d.%s.npy" % (base, b, k)
                    mm = np.lib.format.open_memmap(path, mode="w+", dtype=v.dtype, shape=v.shape)
                    mm[...] = v
                    mm.flush()
                    del mm
                    big = k in _SHARED_KEYS
                    qb[k] = np.load(path, mmap_mode="r") if big else v      # rank 0 reads the shared copy of the images too and drops its private one
                    mb[k] = {"npy": path, "mmap": big}
                else:
                    qb[k] = v
                    mb[k] = v
            meta[str(b)] = mb
            out[b] = qb
        with open(meta_path + ".tmp", "w") as fh:
            json.dump(meta, fh)
        os.replace(meta_path + ".tmp", meta_path)
        open(ready, "w").close()
        open(mapped_marker(0), "w").close()
        t0 = time.time()
        while not all(os.path.exists(mapped_marker(r)) for r in range(local_world)):   # then the names can go: the mappings keep the pages
            if time.time() - t0 > timeout_s:
                break
            time.sleep(0.05)
        for b, mb in meta.items():
            for k, v in mb.items():
                if isinstance(v, dict) and "npy" in v and os.path.exists(v["npy"]):
                    os.unlink(v["npy"])
        for f in [ready, meta_path] + [mapped_marker(r) for r in range(local_world)]:
            if os.path.exists(f):
                os.unlink(f)
        if log:
            log("base streams shared through %s (%d local ranks)" % (shm_dir, local_world))
        return out
    t0 = time.time()
    while not os.path.exists(ready):
        if time.time() - t0 > timeout_s:
            raise TimeoutError("shared base streams: local rank 0 did not publish %s within %.0f s" % (ready, timeout_s))
        time.sleep(0.05)
    with open(meta_path) as fh:
        meta = json.load(fh)
    out = {}
    for b, mb in meta.items():
        q = {}
        for k, v in mb.items():
            q[k] = (np.load(v["npy"], mmap_mode="r") if v["mmap"] else np.load(v["npy"])) if (isinstance(v, dict) and "npy" in v) else v
        out[int(b)] = q
    open(mapped_marker(local_rank), "w").close()
    return out


def _window_totals(systems, per):
    tot = np.zeros(6, np.int64)
    for sy in systems:
        for q in range(sy.S if per is None else per):
            w = sy.lba_window_stats(q)
            tot += np.array([w["windows"], w["local_kfs"], w["fixed_kfs"], w["points"], w["edges"], w["lba_windows_degraded"]], np.int64)
    return tot


def window_stats(after, before):
    """Mean local-BA window of the steps between two `_window_totals` snapshots."""
    d = after - before
    n = max(int(d[0]), 1)
    return {"windows": int(d[0]), "mean_local_kfs": round(float(d[1]) / n, 2), "mean_fixed_kfs": round(float(d[2]) / n, 2), "mean_points": round(float(d[3]) / n, 1),
            "mean_edges": round(float(d[4]) / n, 1), "lba_windows_degraded": int(d[5])}


def run_rank(wl, make_system, rank, world, seqs_per_rank, handles, steps, warmup, on_device, device=None, host_threads=0, collect_poses=False,
             sequences=None, after_warmup=None, coll_on_device=True, preroll=0, post_frames=0, post=None, progress=None, local_mapping=slam.LM_SYNC, total_sequences=None,
             drop_host_inputs=()):
    """Runs this rank's shard: `seqs_per_rank` sequences in `handles` driver handles (each advanced by its own host thread); `preroll` untimed steps that
    bring every sequence's map to its steady state (they are part of the set-up, like loading a map), `warmup` untimed lockstep steps, then exactly `steps`
    timed steps bracketed by a barrier + device synchronisation on both sides.  `post(ctx)` (optional) may run further phases on the warmed sequences
    (`post_frames` more frames are kept per sequence for it).
    Returns (summary dict on every rank, per-rank records [world, len(RECORD_FIELDS)], systems, extra)."""
    import torch
    import torch.distributed as dist
    multi = world > 1 and dist.is_available() and dist.is_initialized()
    # total_sequences (optional): a job whose sequence count is NOT a multiple of the world size (configs[4]: "8 sequences"; any number of sequences on any number
    # of GPUs): rank r still takes sequences r, r + world, ...; the ranks' shards then differ by one sequence and a rank's handles by one as well
    total = seqs_per_rank * world if total_sequences is None else int(total_sequences)
    mine = shard_sequences(total, world, rank)                 # global sequence ids of this rank
    assert len(mine) >= 1, "rank %d of %d has no sequence (total %d)" % (rank, world, total)
    if total_sequences is None:
        assert len(mine) == seqs_per_rank and seqs_per_rank % handles == 0
    seqs_per_rank = len(mine)
    handles = min(handles, seqs_per_rank)
    groups = [[int(x) for x in g] for g in np.array_split(np.asarray(mine), handles)]
    per = None if len({len(g) for g in groups}) > 1 else len(groups[0])
    n_timed0 = preroll + warmup
    n_frames = n_timed0 + steps
    inp = _Inputs(wl, mine, n_frames + post_frames, on_device, device, base_seed=wl.n_base * rank, sequences=sequences)
    if on_device:   # (the images of these base streams are in HBM now; the caller does not need their host copies again: they go back before the maps grow)
        for b in drop_host_inputs:
            for key in ("gray", "right", "depth", "masks"):
                if b in inp.seqs and isinstance(inp.seqs[b].get(key), np.ndarray):   # (a zero-stride stand-in keeps shape and dtype for the code that asks for them)
                    a = inp.seqs[b][key]
                    inp.seqs[b][key] = np.broadcast_to(np.zeros((1,) * a.ndim, a.dtype), a.shape)
    systems = []
    for h in range(handles):
        cfg = slam.make_config(wl.width, wl.height, len(groups[h]), cam=wl.cam, nFeatures=wl.nFeatures, sensor=wl.sensor,
                               device=(device.index if hasattr(device, "index") and device.index is not None else 0) if on_device else 0, host_threads=host_threads,
                               local_mapping=local_mapping)
        systems.append(make_system(cfg))
    poses = [[] for _ in range(handles)] if collect_poses else [None] * handles
    calls = [_prepare(systems[h], wl, inp, groups[h], n_frames + post_frames) for h in range(handles)]

    def sync():
        if on_device:
            torch.cuda.synchronize()

    def phase(t0, t1, calls_=None):
        cl = calls_ if calls_ is not None else calls
        ths = [threading.Thread(target=_drive, args=(systems[h], wl, cl[h], t0, t1, poses[h])) for h in range(handles)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    t_pre = time.perf_counter()
    for t0 in range(0, preroll, 25):      # in chunks, so a long pre-roll reports progress
        phase(t0, min(preroll, t0 + 25))
        if progress is not None:
            progress("pre-roll %d / %d steps, %.1f s" % (min(preroll, t0 + 25), preroll, time.perf_counter() - t_pre))
    sync()
    t_pre = time.perf_counter() - t_pre
    phase(preroll, n_timed0)
    sync()
    if after_warmup is not None:
        after_warmup(systems)
    win0 = _window_totals(systems, per)
    if multi:
        dist.barrier()
    sync()
    tstart = time.perf_counter()
    phase(n_timed0, n_frames)
    sync()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - tstart
    win1 = _window_totals(systems, per)
    frames = seqs_per_rank * steps
    total_frames, max_elapsed = aggregate_stats(elapsed, frames, device=device if (on_device and coll_on_device) else None)

    # per-rank record (fixed size, all-gathered once): counters summed over the rank's sequences, ATE of its first sequence
    kf = lba = lost = viol = sem = 0
    for h in range(handles):
        for q in range(len(groups[h])):
            st = systems[h].stats(q)
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
               "ate_rmse_m": float(np.nanmean(records[:, 7])), "map_violations": int(records[:, 8].sum()), "semantic_edges": int(records[:, 9].sum()),
               "preroll_steps": preroll, "preroll_s": t_pre, "lba_windows_timed": window_stats(win1, win0), "lba_windows_all": window_stats(win1, np.zeros(6, np.int64))}
    extra = {"inputs": inp, "groups": groups, "poses": poses, "calls": calls, "phase": phase, "sync": sync, "n_frames": n_frames, "per": per}
    if post is not None:
        extra["post"] = post(dict(extra, systems=systems, wl=wl, window_totals=lambda: _window_totals(systems, per)))
    return summary, records, systems, extra


def pack_mask_bits(masks):
    """uint8 masks [..., H, W] ({0, 255}) -> one bit per pixel, uint64 [..., H, ceil(W / 64)] in the layout of oslam_mask_bits_device (bit i of word w = pixel 64 w + i == 255)."""
    H, W = masks.shape[-2:]
    WB = (W + 63) // 64
    b = np.zeros(masks.shape[:-1] + (WB * 64,), np.uint8)
    b[..., :W] = masks == 255
    return np.ascontiguousarray(np.packbits(b, axis=-1, bitorder="little")).view("<u8")


def host_input_calls(ctx, first, n):
    """Call arguments of steps [first, first + n) of every handle with the inputs in PINNED HOST memory in the raw formats (8-bit gray, 16-bit depth with
    DepthMapFactor 5000, instance masks as one bit per pixel): oslam_slam_track_rgbd_raw16 with device-accessible pointers — the kernels read the pinned
    buffers over PCIe, nothing is staged.  Returns (per-handle call lists indexed by absolute step, input bytes per frame)."""
    import torch
    from .scene import DEPTH_FACTOR, depth_to_metres
    inp, wl, systems, groups = ctx["inputs"], ctx["wl"], ctx["systems"], ctx["groups"]
    last = first + n + wl.stagger
    pinned = {}
    for b, q in inp.seqs.items():
        d16 = np.rint(q["depth"][first:last].astype(np.float64) * DEPTH_FACTOR).astype(np.uint16)
        assert np.array_equal(depth_to_metres(d16), q["depth"][first:last])      # the float images of the HBM leg are exactly these raw images scaled
        bits = pack_mask_bits(q["masks"][first:last])
        pinned[b] = {"gray": torch.from_numpy(np.ascontiguousarray(q["gray"][first:last])).pin_memory(), "depth": torch.from_numpy(d16).pin_memory(),
                     "masks": torch.from_numpy(bits.view(np.int64)).pin_memory()}
    nobj = next(iter(inp.seqs.values()))["masks"].shape[1]

    def table(seq_ids, key, sub=None):
        S = len(seq_ids)
        out = np.zeros((n, S), np.uint64)
        for i, g in enumerate(seq_ids):
            t = pinned[inp.base[g]][key]
            a = t.data_ptr() + (sub * t.stride(1) * t.element_size() if sub is not None else 0)
            out[:, i] = a + (np.arange(n, dtype=np.uint64) + np.uint64(inp.off[g])) * np.uint64(t.stride(0) * t.element_size())
        return out

    calls = []
    for h, sy in enumerate(systems):
        ids = groups[h]
        S = len(ids)
        stamps = np.repeat(((first + np.arange(n, dtype=np.float64)) / wl.fps)[:, None], S, 1)
        masks = np.stack([table(ids, "masks", o) for o in range(nobj)], 2)
        tids = np.array([inp.seqs[inp.base[g]]["track_ids"] for g in ids], np.int32)
        labs = np.array([inp.seqs[inp.base[g]].get("labels", [0] * nobj) for g in ids], np.int32)
        cl = sy.prepare_rgbd_bulk(table(ids, "gray"), table(ids, "depth"), stamps, wl.width, wl.width, masks=masks, track_ids=tids, labels=labs,
                                  depth_u16_factor=float(np.float32(1.0) / np.float32(DEPTH_FACTOR)), mask_bits=True, on_device=1)
        calls.append([None] * first + cl)
    ctx["_pinned"] = pinned
    WB = (wl.width + 63) // 64
    return calls, wl.width * wl.height * 3 + nobj * wl.height * WB * 8


# ---- the two stream shapes of SURVEY.md §8(d) ----
def _make_rgbd(seed, n, speed=1.0, with_masks=True, first=0, count=None):
    from . import scene
    return scene.make_rgbd_sequence(seed, n, speed=speed, with_masks=with_masks, first=first, count=count)


def _make_stereo(seed, n, speed=0.35, first=0, count=None):
    from . import scene
    return scene.make_stereo_sequence(seed, n, speed=speed, first=first, count=count)


def rgbd_workload(speed=1.0, with_masks=True, n_base=8, stagger=8):
    """S1: TUM-shaped RGB-D room with three box objects (640x480, 1000 features, TUM2.yaml calibration)."""
    import functools
    return Workload("S1 TUM-shaped RGB-D", 640, 480, slam.RGBD, slam.TUM2, 1000, 30.0, functools.partial(_make_rgbd, speed=speed, with_masks=with_masks), n_base, stagger)


def stereo_workload(speed=0.35, n_base=4, stagger=4):
    """S3/S4: KITTI-shaped rectified stereo street (1241x376, 2000 features, KITTI00-02.yaml calibration)."""
    import functools
    return Workload("S3 KITTI-shaped stereo", 1241, 376, slam.STEREO, slam.KITTI00, 2000, 10.0, functools.partial(_make_stereo, speed=speed), n_base, stagger)
