#!/usr/bin/env python3
"""bench.py — frames/s of tracking + local BA (BASELINE.json `metric`) through the MI355X-native batch-of-sequences driver.

Workload (all N): S1, the TUM-shaped synthetic RGB-D stream of SURVEY.md §8(d) (640x480, 1000 ORB features, 3 object masks, SE3 path with
rotation) — `--workload stereo` switches the headline to S3/S4 (KITTI-shaped stereo, 1241x376, 2000 features, BASELINE.json configs[3]/[4]),
which is otherwise reported as the second figure `stereo`.  A "step" is one frame of every sequence of the rank: Frame construction
(ORBextractor ...), TrackWithMotionModel / TrackReferenceKeyFrame, TrackLocalMap, keyframe decision and — for the sequences that inserted
a keyframe — one pass of LocalMapping::Run including LocalBundleAdjustment (reference timing points: Examples/RGB-D/rgbd_tum.cc:93-109 around
the track call, src/LocalMapping.cc:82).  Images are resident in HBM when the timed region starts.

Contract: python bench.py --gpus N --steps K --warmup W ; ONE JSON line on rank 0.  `--gpus N` with no RANK in the environment starts
the N ranks itself (child processes, before anything touches the GPU); under torchrun the RANK/LOCAL_RANK/WORLD_SIZE of the
environment are used.  Multi-GPU = independent sequences sharded over ranks (weak scaling: `--seqs` sequences per GPU), no data-path
collective; RCCL carries the barriers, the max-over-ranks time and one all-gather of a fixed-size record per rank.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# One hardware queue per driver handle.  HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES queues (default 4) and a queue executes its packets in
# order whatever stream they came from: with 8 handles, two handles shared a queue and the short kernels of one waited behind the ~180 local-BA launches of the
# other (the MapPoint-update operator spent 3.5x its kernels' time waiting).  Must be in the environment before the HIP runtime starts (the first import of
# torch): measured 24.2 k against 22.7 k frames/s in the steady state (same box); 12 / 16 queues give 24.2 / 24.4 k.  INTEGRATION.md names it for host applications.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")   # (round 4: 8 handles + the 2 streams of the local-BA service; same-box A/B of the deferred schedule: 8 -> 26.4 k, 12 -> 27.3 k, 16 -> 26.6 k)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 matrix (= vector) peak, AMD datasheet: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64: 2048 flop / 64 clk;
                           # MI355X_MICROARCH.md lists no fp64 row; tools/mfma_f64_rate.py measures the issue rate on the box)


# --------------------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torchrun environment
# --------------------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """Start n rank processes of this script (one per GPU) and relay rank 0's stdout.  The parent never touches the GPU."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


# --------------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the same driver over the CPU oracle's operator table (test infrastructure; "port"), bounded sample
# --------------------------------------------------------------------------------------------------------------------------------
def extend_sequence(wl, seed, seq, extra, workers=1, chunk=12):
    """`seq` (frames [0, n) of stream `seed`) continued by `extra` frames of the SAME stream (the paths of object_slam_amd/scene.py are functions of the frame
    index alone, so frames [n, n + extra) rendered later are the frames a longer rendering would have held).  A copy: the GPU legs keep their arrays."""
    n = len(seq["gray"])
    jobs = [(wl, seed, n + extra, f, min(chunk, n + extra - f)) for f in range(n, n + extra, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.starmap(_render_piece, jobs)
    else:
        parts = [_render_piece(*j) for j in jobs]
    out = dict(seq)
    T0 = seq["Twc"][0]
    for key in ("gray", "right", "depth", "masks"):
        if seq.get(key) is not None:
            out[key] = np.concatenate([seq[key]] + [p[key] for p in parts], 0)
    out["Twc"] = np.concatenate([seq["Twc"]] + [p["Twc"] for p in parts], 0)   # (every piece is relative to frame 0 of the whole stream)
    assert np.allclose(out["Twc"][0], T0)
    return out


def _render_piece(wl, seed, n, first, count):
    return wl.make_sequence(seed, n, first=first, count=count)


def _cpu_stream(L, make_ops, wl, seq, first, n_timed, local_mapping):
    """One base stream through the product's driver over the CPU oracle's operator table `make_ops` of library L: `first` frames untimed, then `n_timed` timed."""
    import ctypes as C
    from object_slam_amd import slam
    from object_slam_amd.io import horn_align_ate
    cfg = slam.make_config(wl.width, wl.height, 1, cam=wl.cam, nFeatures=wl.nFeatures, sensor=wl.sensor, local_mapping=local_mapping)
    ops = slam.SlamOps()
    assert getattr(L, make_ops)(C.byref(cfg), C.byref(ops)) == 0
    sysm = slam.System(cfg, ops)
    n = min(first + n_timed, len(seq["gray"]))
    has_masks = seq.get("masks") is not None

    def step(t):
        if wl.sensor == slam.STEREO:
            sysm.TrackStereo([seq["gray"][t]], [seq["right"][t]], [t / wl.fps])
        else:
            objs = [dict(masks=[seq["masks"][t, o] for o in range(seq["masks"].shape[1])], track_ids=seq["track_ids"], labels=seq.get("labels"))] if has_masks else None
            if objs is not None and objs[0]["labels"] is None:
                objs[0].pop("labels")
            sysm.TrackRGBD([seq["gray"][t]], [seq["depth"][t]], [t / wl.fps], objects=objs)

    tp = time.perf_counter()
    for t in range(first):
        step(t)
    if hasattr(sysm, "finish"):
        sysm.finish()
    tp = time.perf_counter() - tp
    w0 = sysm.lba_window_stats(0)
    stg0 = sysm.stage_seconds()
    per = []
    t0 = time.perf_counter()
    for t in range(first, n):
        t1 = time.perf_counter()
        step(t)
        per.append(time.perf_counter() - t1)
    if hasattr(sysm, "finish"):
        sysm.finish()          # (deferred schedule: the local BA in flight belongs to the timed frames)
    dt = time.perf_counter() - t0
    w1 = sysm.lba_window_stats(0)
    stg1 = sysm.stage_seconds()
    map_s = sum(stg1.get(k, 0.0) - stg0.get(k, 0.0) for k in ("lba", "host_mapping", "fuse_bow_triangulate", "mp_update"))
    _, Twc = sysm.trajectory(0)
    T0inv = np.linalg.inv(seq["Twc"][0])
    gt = np.array([T0inv @ x for x in seq["Twc"][:len(Twc)]])
    st = sysm.stats(0)
    out = {"frames": n - first, "dt": dt, "preroll_s": tp, "map_s": map_s, "per": per, "ate": horn_align_ate(Twc[:, :, 3], gt[:, :3, 3]),
           "keyframes": st["keyframes_created"], "local_bas": st["local_bas"],
           "win": np.array([w1[k] - w0[k] for k in ("windows", "local_kfs", "fixed_kfs", "points", "edges")], np.float64)}
    sysm.close()
    return out


def cpu_baseline(wl, seqs, first, n_timed, local_mapping=0x1F, two_thread_streams=4):
    """The same driver over the CPU oracle's operator table (kind "port"; built -march=native on this host as BASELINE.md section 3 asks), on EVERY base stream in
    `seqs` over the stream frames the GPU leg's sequences cover in its timed region (`first` frames untimed so that the map has the same age, then `n_timed` frames
    timed).  The streams run side by side, one core each (`value` = timed frames / core-seconds: the one-core rate); then the reference's threading shape is
    MEASURED on `two_thread_streams` of them: the deferred schedule with the local BA on a second thread (src/System.cc:95), two cores per stream."""
    import threading
    from object_slam_amd import slam
    from oracle import oracle_py as O
    O.build()
    L, march = O.native_lib()

    def run_all(make_ops, lm, which):
        res = [None] * len(which)

        def work(i):
            res[i] = _cpu_stream(L, make_ops, wl, seqs[which[i]], first, n_timed, lm)
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(which))]   # (the operators release the GIL: ctypes calls)
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return res, time.perf_counter() - t0

    which = sorted(seqs)[:8]
    res, wall = run_all("oo_slam_make_ops", local_mapping, which)
    frames = sum(r["frames"] for r in res)
    core_s = sum(r["dt"] for r in res)
    per = np.concatenate([np.array(r["per"]) for r in res])
    win = sum(r["win"] for r in res)
    nw = max(1.0, win[0])
    hidden = sum(r["frames"] for r in res) / max(sum(r["dt"] - r["map_s"] for r in res), 1e-9)
    two = None
    if two_thread_streams > 0 and hasattr(L, "oo_slam_make_ops_threaded"):
        w2 = which[:two_thread_streams]
        r2, wall2 = run_all("oo_slam_make_ops_threaded", slam.LM_DEFERRED, w2)
        per2 = np.concatenate([np.array(r["per"]) for r in r2])
        two = {"value": round(sum(r["frames"] for r in r2) / sum(r["dt"] for r in r2), 2), "unit": "frames/s per stream", "cores": 2, "streams": len(w2),
               "timed_frames": int(sum(r["frames"] for r in r2)), "mean_ms_per_frame": round(float(per2.mean()) * 1e3, 2), "median_ms_per_frame": round(float(np.median(per2)) * 1e3, 2),
               "local_bas": int(sum(r["local_bas"] for r in r2)), "ate_rmse_m": round(float(np.mean([r["ate"] for r in r2])), 6),
               "shape": "tracking (with the rest of the local-mapping pass) on one thread, Optimizer::LocalBundleAdjustment of keyframe t on a second thread while frame t + 1 is "
                        "tracked (deferred schedule, oo_slam_make_ops_threaded): the reference's LocalMapping thread, src/System.cc:95"}
    # the reference prints median and mean tracking time per frame (Examples/RGB-D/rgbd_tum.cc:126-134)
    return {"value": round(frames / core_s, 2), "unit": "frames/s", "cores": 1, "kind": "port", "march": march,
            "sample": "stream frames %d..%d of the %d base streams — the frames the GPU leg's sequences (offsets 0 .. stagger in their base streams) cover in its timed region, each "
                      "with a map of the same age (%d untimed frames first) — through the same driver over the CPU oracle's operator table; the streams run side by side on "
                      "%d CPUs, tracking AND local mapping of a stream on ONE core one after the other; %.1f core-seconds timed (%.1f s wall incl. %.1f s of untimed pre-roll per stream)"
                      % (first, first + n_timed, len(which), first, len(which), core_s, wall, float(np.mean([r["preroll_s"] for r in res]))),
            "streams": len(which), "frames_per_s_by_stream": [round(r["frames"] / r["dt"], 2) for r in res],
            # the reference's shape (<= 3 busy cores: LocalMapping on a second thread, src/System.cc:95; stereo extraction on two, src/Frame.cc:78-81): its frames/s lies between
            # the one-core figure and the figure with every mapping stage hidden behind tracking; `two_thread` is the measured run of that shape
            "three_core_bracket_frames_per_s": [round(frames / core_s, 2), round(hidden, 2)], "two_thread": two,
            "mapping_stage_seconds": round(sum(r["map_s"] for r in res), 2), "timed_frames": int(frames),
            "mean_ms_per_frame": round(float(per.mean()) * 1e3, 2), "median_ms_per_frame": round(float(np.median(per)) * 1e3, 2),
            "ate_rmse_m": round(float(np.mean([r["ate"] for r in res])), 6), "keyframes": int(sum(r["keyframes"] for r in res)), "local_bas": int(sum(r["local_bas"] for r in res)),
            "lba_windows_timed": {"windows": int(win[0]), "mean_local_kfs": round(win[1] / nw, 2), "mean_fixed_kfs": round(win[2] / nw, 2), "mean_points": round(win[3] / nw, 1),
                                  "mean_edges": round(win[4] / nw, 1)}}


def single_sequence(wl, seq, first, n_timed, device_index):
    """The HIP path as ONE caller sees it (the reference's only timing: median / mean tracking time per frame of one sequence, Examples/RGB-D/rgbd_tum.cc:93-134):
    one sequence, host images handed over per frame, the same stream frames the CPU port is timed on, `first` frames untimed."""
    from object_slam_amd import slam
    sysm = slam.System(slam.make_config(wl.width, wl.height, 1, cam=wl.cam, nFeatures=wl.nFeatures, sensor=wl.sensor, host_threads=1, device=device_index))
    has_masks = seq.get("masks") is not None
    n = min(first + n_timed, len(seq["gray"]))
    per = []
    for t in range(n):
        t1 = time.perf_counter()
        if wl.sensor == slam.STEREO:
            sysm.TrackStereo([seq["gray"][t]], [seq["right"][t]], [t / wl.fps])
        else:
            objs = [dict(masks=[seq["masks"][t, o] for o in range(seq["masks"].shape[1])], track_ids=seq["track_ids"])] if has_masks else None
            sysm.TrackRGBD([seq["gray"][t]], [seq["depth"][t]], [t / wl.fps], objects=objs)
        if t >= first:
            per.append(time.perf_counter() - t1)
    st = sysm.stats(0)
    sysm.close()
    per = np.array(per) * 1e3
    return {"sequences": 1, "timed_frames": int(len(per)), "mean_ms_per_frame": round(float(per.mean()), 3), "median_ms_per_frame": round(float(np.median(per)), 3),
            "p95_ms_per_frame": round(float(np.percentile(per, 95)), 3), "frames_per_s": round(1e3 / float(per.mean()), 1), "local_bas": st["local_bas"], "lost_frames": st["lost_frames"],
            "note": "one sequence through the oslam_slam driver over the HIP operator table, host images uploaded per frame (PCIe inside the figure), local mapping incl. local BA "
                    "inside the frame call that inserted the keyframe: latency, not throughput — the card is idle most of each frame"}


# --------------------------------------------------------------------------------------------------------------------------------
# S2 stage entry: ORBextractor + ORBmatcher only (BASELINE.json configs[1]), per-kernel HBM table
# --------------------------------------------------------------------------------------------------------------------------------
def lba_alone(device_index, windows=40, reps=3, mode=1):
    """The local-BA operator ALONE on the card: one call of `windows` steady-state-shaped windows (27 free keyframes, 1500 points, ~13 k edges: the shape of the
    timed steps' windows, synthetic geometry of object_slam_amd/synth.py), multi-launch layout as the driver uses it; device time from the handle's HIP events,
    fp64 work = the SURVEY.md §8(d) flop model x the LM trials the kernels report.  The headline's local-BA figure is the same kernels under the contention of 8
    handles; this is what they reach by themselves."""
    import ctypes as C
    from object_slam_amd import LocalBundleAdjuster, synth
    base = [synth.make_lba_problem(1234 + i, K_local=27, K_fixed=0, P=1500, track=13, stereo_frac=0.9) for i in range(8)]
    probs = [base[i % len(base)] for i in range(windows)]
    ba = LocalBundleAdjuster(max_batch=windows, max_keyframes=64, max_points=8192, max_edges=65536, device=device_index)
    ba.set_mode(mode)
    try:
        ms, ln = C.c_double(0), C.c_longlong(0)
        out = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])            # warm (allocations)
        ba.L.oslam_lba_kernel_time(ba.h, 1, C.byref(ms), C.byref(ln))       # enable + reset
        for _ in range(reps):
            out = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
        ba.L.oslam_lba_kernel_time(ba.h, 1, C.byref(ms), C.byref(ln))
        flop = 0.0
        for q, o in zip(probs, out):
            k = np.bincount(np.asarray(q["edge_pt"]), minlength=len(q["points"])).astype(np.float64)
            n6 = 6.0 * int((np.asarray(q["fixed"]) == 0).sum())
            flop += (o[3][1] + o[3][3]) * (700.0 * len(q["edge_kf"]) + 324.0 * float((k * k).sum()) + n6 ** 3 / 3.0 + 2.0 * n6 * n6 + 45.0 * len(q["points"]))
        ms_call = ms.value / reps
        tf = flop / (ms_call * 1e-3) / 1e12
        return {"windows": windows, "layout": "multi-launch (mode 1)" if mode == 1 else "one workgroup per window (mode 2)",
                "edges_per_window": int(len(probs[0]["edge_kf"])), "ms_per_call": round(ms_call, 3), "launches_per_call": int(ln.value // reps),
                "fp64_TFLOPs": round(tf, 3), "frac_of_fp64_peak": round(tf / FP64_PEAK_TFLOPS, 4)}
    finally:
        ba.close()


def frontend_stage(frames, Twc, depth, local_rank, steps, B=512, parts=None):
    """Batched extraction of B frames + SearchByProjection(Cur, Last) with the ground-truth pose, everything resident in HBM.
    The B frames of a step are cut into `parts` sub-batches, each with its own extractor / matcher handle and HIP stream: the quad-tree, descriptor
    and matching kernels of one sub-batch (latency bound, few wavefronts) then run beside the FAST / blur kernels (VALU bound) of the others, the
    way the driver's handles overlap.  parts = 1 is the single-stream form of round 1."""
    import ctypes as C

    import torch

    from object_slam_amd import ORBextractor, ORBmatcher, slam
    from object_slam_amd._lib import check
    from object_slam_amd.matcher import MatchFrames, MatchLast
    parts = parts or int(os.environ.get("OSLAM_S2_PARTS", "1"))   # measured on one MI355X: 1 -> 171 k, 2 -> 176 k, 4 -> 128 k, 8 -> 145 k frames/s (tools/frontend_parts.py)
    assert B % parts == 0
    n_src, H, W = frames.shape
    NFEAT, NLEVELS, TH = 1000, 8, 15.0
    cam5 = slam.TUM2
    cam = (cam5["fx"], cam5["fy"], cam5["cx"], cam5["cy"], cam5["bf"], cam5["bf"] / cam5["fx"])
    idx_all = np.arange(B) % (n_src - 1) + 1                 # frame b = source frame idx[b], its "last frame" = idx[b] - 1
    pitch = (W + 63) // 64 * 64
    Bp = B // parts

    class Part:
        pass

    def make_part(k):
        q = Part()
        idx = idx_all[k * Bp:(k + 1) * Bp]
        q.stream = torch.cuda.Stream() if parts > 1 else torch.cuda.current_stream()
        st = q.st = q.stream.cuda_stream
        q.d_img = torch.zeros((Bp, H, pitch), dtype=torch.uint8, device="cuda")
        q.d_img[:, :, :W] = torch.from_numpy(frames[idx]).cuda()
        ex = q.ex = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, W, H, max_batch=Bp, device=local_rank)
        cap = q.cap = ex.cap
        q.mt = ORBmatcher(0.9, True, max_keypoints=cap, max_queries=cap, max_batch=Bp, device=local_rank)
        q.sf = ex.GetScaleFactors()
        d_kp, d_desc, d_cnt, _ = ex.results_device()
        # "last frames": extraction of the previous source frames, map points from their depth
        d_prev = torch.zeros((Bp, H, pitch), dtype=torch.uint8, device="cuda")
        d_prev[:, :, :W] = torch.from_numpy(frames[idx - 1]).cuda()
        torch.cuda.synchronize()
        ex.extract_batch_device(d_prev.data_ptr(), Bp, pitch, pitch * H, st)
        torch.cuda.synchronize()
        host = [ex.fetch(b) for b in range(Bp)]
        last_keys = np.zeros((Bp, cap), dtype=host[0][0].dtype)
        last_desc = np.zeros((Bp, cap, 32), np.uint8)
        last_Xw = np.zeros((Bp, cap, 3), np.float32)
        last_has = np.zeros((Bp, cap), np.uint8)
        q.last_n = last_n = np.zeros(Bp, np.int32)
        Tcw = np.zeros((Bp, 4, 4), np.float32)
        Tlw = np.zeros((Bp, 4, 4), np.float32)
        for b in range(Bp):
            kl, dl = host[b]
            n = len(kl)
            Tl = np.linalg.inv(Twc[idx[b] - 1])
            z = depth[idx[b] - 1][kl["y"].astype(np.int64), kl["x"].astype(np.int64)].astype(np.float64)
            Xc = np.stack([(kl["x"] - cam[2]) * z / cam[0], (kl["y"] - cam[3]) * z / cam[1], z], 1)
            Xw = (Xc - Tl[:3, 3]) @ Tl[:3, :3]
            last_keys[b, :n], last_desc[b, :n], last_Xw[b, :n], last_n[b] = kl, dl, Xw, n
            last_has[b, :n] = np.where(z > 0, 3, 0)
            Tcw[b], Tlw[b] = np.linalg.inv(Twc[idx[b]]), Tl
        q.keep = [torch.from_numpy(last_keys.view(np.uint8).reshape(Bp, -1)).cuda()] + [torch.from_numpy(a).cuda() for a in (last_desc, last_Xw, last_has, last_n, Tcw, Tlw)]
        t_keys, t_desc, t_Xw, t_has, t_n, q.t_Tcw, q.t_Tlw = q.keep
        q.t_uR = torch.full((Bp, cap), -1.0, dtype=torch.float32, device="cuda")
        fr = q.fr = MatchFrames()
        fr.keysUn, fr.kp_stride, fr.uRight, fr.desc, fr.blocked = d_kp, cap, q.t_uR.data_ptr(), d_desc, None
        fr.n_kps, fr.n_kps_const = d_cnt, 0
        fr.minX, fr.minY, fr.maxX, fr.maxY = 0.0, 0.0, float(W), float(H)
        la = q.la = MatchLast()
        la.Xw, la.has_mp, la.keys, la.mp_desc = t_Xw.data_ptr(), t_has.data_ptr(), t_keys.data_ptr(), t_desc.data_ptr()
        la.kp_stride, la.n_kps, la.n_kps_const = cap, t_n.data_ptr(), 0
        q.q_nq = C.c_void_p()
        check(q.mt.L.oslam_match_results_device(q.mt.h, None, None, None, None, None, C.byref(q.q_nq)))
        return q

    P = [make_part(k) for k in range(parts)]
    torch.cuda.synchronize()
    ex, mt, cap, last_n = P[0].ex, P[0].mt, P[0].cap, P[0].last_n
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step_part(q, timed_match=False):
        q.ex.extract_batch_device(q.d_img.data_ptr(), Bp, pitch, pitch * H, q.st)
        if timed_match:
            ev0.record(q.stream)
        q.mt.project_last_batch_device(q.la, q.t_Tcw.data_ptr(), q.t_Tlw.data_ptr(), cam, q.fr, q.sf, TH, False, Bp, q.st)
        q.mt.search_batch_device(q.fr, None, q.cap, q.q_nq.value, 0, Bp, False, True, q.st)
        if timed_match:
            ev1.record(q.stream)

    def step():
        for q in P:
            step_part(q)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    counts = np.array([len(q.ex.fetch(b)[0]) for q in P for b in range(0, Bp, 37)])
    nm = mt.fetch(Bp // 2, cap, int(last_n[Bp // 2]), cap, len(ex.fetch(Bp // 2)[0]), P[0].st)[0]
    # per-kernel table: sub-batch 0 alone (its kernels then have the GPU to themselves, as in the single-stream form)
    ex.set_profiling(1)
    match_ms = []
    for _ in range(5):
        step_part(P[0], timed_match=True)
        torch.cuda.synchronize()
        match_ms.append(ev0.elapsed_time(ev1))
    ms, nb, ni = ex.get_profile()
    ex.set_profiling(0)
    n_kp = float(counts.mean())
    Ptot = sum(ex.level_size(l)[0] * ex.level_size(l)[1] for l in range(NLEVELS))
    p0, pl = W * H, ex.level_size(NLEVELS - 1)[0] * ex.level_size(NLEVELS - 1)[1]
    alg = {"pyramid(K1)": (Ptot - pl) + (Ptot - p0), "fast_cells(K2/K3)": Ptot, "blur(K6)": 2 * Ptot, "octree(K4)": 0,
           "orient_describe(K5/K7)": (749 + 512 + 60) * n_kp}
    us = {n: m / ni * 1e3 for n, m in zip(alg.keys(), ms)}
    us["match(K8-K10)"] = float(np.mean(match_ms)) / Bp * 1e3
    alg["match(K8-K10)"] = 44 * n_kp + 36 * n_kp + 12288 + 8 * n_kp
    pairs = C.c_int64(0)
    check(mt.L.oslam_match_hamming_pairs(mt.h, Bp, C.byref(pairs)))
    out = {"workload": "S2 (BASELINE.json configs[1]): %d frames of the S1 stream per step in %d sub-batches on their own streams, ORBextractor + SearchByProjection(Cur, Last) "
                       "with the ground-truth pose; the per-kernel table is one sub-batch of %d frames alone" % (B, parts, Bp),
           "parts": parts,
           "frames_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "keypoints_per_frame": round(n_kp, 1), "matches_frame_mid": int(nm),
           "per_frame_us": {k: round(v, 3) for k, v in us.items()}, "alg_bytes_per_frame": {k: int(v) for k, v in alg.items()},
           "GBs": {k: round(alg[k] / (us[k] * 1e-6) / 1e9, 1) for k in us if us[k] > 0},
           "whole_path_GBs": round(ex.algorithmic_bytes(int(n_kp)) * B * steps / dt / 1e9, 1),
           "hamming_pairs_per_frame": round(pairs.value / Bp, 1),
           # committed PMC passes of this stage (profiles/r05_pmc_traffic.json): HBM bytes per frame and the VALU issue share of the kernels' own duration
           "pmc_hbm_KB_per_frame": {k: round(_pmc(k) / 512 / 1e3, 1) for k in ("k_resize_lds", "k_fast_cells_wave", "k_blur_strip<false>", "k_blur_strip<true>", "k_octree",
                                                                               "k_orient_describe", "k_search_window") if _pmc(k) is not None},
           "pmc_valu_issue_frac": {k: _pmc(k, "valu_issue_frac_at_4_cycles") for k in ("k_resize_lds", "k_fast_cells_wave", "k_blur_strip<false>", "k_octree", "k_orient_describe",
                                                                                       "k_search_window") if _pmc(k, "valu_issue_frac_at_4_cycles") is not None},
           "bound_note": "k_fast_cells_wave (70 %) and k_blur_strip (27 %) run concurrently and together saturate the VALU issue: the front-end phase is instruction bound, "
                         "its GB/s figures are reported against the HBM roofline for completeness"}
    for q in P:
        q.ex.close(); q.mt.close()
    return out


def _pmc(kernel, field="bytes_per_launch"):
    """Per-launch PMC figure of `kernel` from the committed summary of this round (profiles/r05_pmc_traffic.json: separate rocprofv3 --pmc passes of
    the S2 stage at 512 frames per launch — FETCH_SIZE, WRITE_SIZE, SQ_* — corrected as MI355X_MICROARCH.md §HBM prescribes); None when the summary has
    no row for it.  HBM bytes for the fp64 solver kernels were not collected (their bound is not HBM)."""
    path = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")   # (re-taken in round 5: tools/gpu/prof_stereo.sh; the front-end kernels are those of round 2)
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(kernel, {}).get(field)
    except Exception:
        return None


def physical_cores():
    """The CPUs this process may run on, grouped by physical core (SMT siblings together), cores in (package, core) order."""
    allowed = sorted(os.sched_getaffinity(0))
    groups = {}
    for c in allowed:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            key = (int(open(base + "physical_package_id").read()), int(open(base + "core_id").read()))
        except Exception:
            key = (0, c)
        groups.setdefault(key, []).append(c)
    return [groups[k] for k in sorted(groups)]


def _package_of(cpu):
    try:
        return int(open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % cpu).read())
    except Exception:
        return 0


def pin_rank_cpus(local_rank, local_world, per_rank_cores):
    """Gives every rank of a multi-rank run a disjoint, contiguous set of physical cores (both SMT threads of a core go to the same rank; contiguous cores
    share a socket / NUMA node) — before the process touches the GPU or starts a thread, so the render pool, the driver's workers and torch inherit it.
    A single rank is kept on one socket.  Returns the number of CPUs the rank may use."""
    cores = physical_cores()
    if local_world <= 1:
        # one rank on the box: keep its threads (and, by first touch, the maps they build) on ONE socket.  Measured in the steady state on a two-socket
        # box with the affinity of either NUMA node: 17.35 k / 17.37 k frames/s against 16.5-16.7 k unpinned (OSLAM_BENCH_NO_PIN=1 leaves the rank unpinned).
        packages = sorted({_package_of(g[0]) for g in cores})
        if len(packages) > 1 and not os.environ.get("OSLAM_BENCH_NO_PIN"):
            cpus = sorted(c for g in cores if _package_of(g[0]) == packages[0] for c in g)
            os.sched_setaffinity(0, cpus)
            return len(cpus)
        return sum(len(c) for c in cores)
    cpus = rank_cpu_set(cores, local_rank, local_world, per_rank_cores)
    os.sched_setaffinity(0, cpus)
    return len(cpus)


def rank_cpu_set(cores, local_rank, local_world, per_rank_cores=0):
    """The CPUs of rank `local_rank` of `local_world` ranks on a node whose physical cores are `cores` (lists of SMT siblings in (package, core) order): a
    contiguous run of len(cores) // local_world cores (optionally only the first per_rank_cores of them), both SMT threads of a core to the same rank."""
    per = max(1, len(cores) // local_world)
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores[-per:]
    if per_rank_cores:
        mine = mine[:max(1, per_rank_cores)]
    return sorted(c for g in mine for c in g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("rgbd", "stereo"), default="rgbd")
    ap.add_argument("--preroll", type=int, default=-1, help="untimed set-up steps before the warm-up that bring every sequence's map to its steady state (default: 200 for the "
                                                             "RGB-D stream = SURVEY.md §8(d) frame >= 200, 40 for the stereo street); 0 = the cold-start regime of round 2")
    ap.add_argument("--seqs", type=int, default=0, help="sequences per GPU (default: 8192 RGB-D / 2048 stereo)")
    ap.add_argument("--bases", type=int, default=8, help="base renderings per GPU of the headline stream (distinct input streams = bases x 25 frame offsets)")
    ap.add_argument("--lm", choices=("deferred", "sync"), default="sync", help="local-mapping schedule (include/oslam_slam.h): sync (default) = the whole pass right after the frame "
                    "that inserted the keyframe, its local BA solved by the process-wide service in batches shared with the other handles; deferred = the local BA of keyframe "
                    "t is solved while frame t+1 is tracked, its write-back and KeyFrameCulling land before frame t+2 (the reference's two-thread overlap, src/System.cc:95).  "
                    "On the final code of round 4 sync is the faster one on the headline stream (12 %% fewer keyframes, DESIGN.md section 7.2)")
    ap.add_argument("--handles", type=int, default=0, help="driver handles per GPU, one host thread each (default 8)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline's timed range (default: the timed steps and what the base sequence holds after them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--strict", action="store_true", help="exit non-zero when one of the extra legs (front end, local BA alone, host inputs, second workload ...) failed, instead of "
                    "only recording its error in the line")
    ap.add_argument("--no-bases32", action="store_true", help="skip the second headline figure with 32 base renderings (800 distinct input streams)")
    ap.add_argument("--extras-child-processes", action="store_true", help="run the second workload and the 32-base leg each in a child process of this program instead of "
                    "inside this process behind the headline (measured in round 5: much slower while the parent process is alive on the card - 4.5 k against 18-22 k stereo frames/s)")
    ap.add_argument("--cold", action="store_true", help="also run the cold-start regime of rounds 1-2 (steps W..W+K of empty maps) on the headline streams")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # a disjoint CPU set per rank, before anything forks, starts a thread or touches the GPU
    ncpu = pin_rank_cpus(local_rank, local_world, int(os.environ.get("OSLAM_BENCH_CORES_PER_RANK", "0")))
    share = min(16, ncpu)        # host threads of this rank: 16 = the box's CPU share per GPU

    from object_slam_amd import seqbench, slam
    cores = os.cpu_count() or 1
    stereo_head = args.workload == "stereo"
    preroll = args.preroll if args.preroll >= 0 else (40 if stereo_head else 200)
    wl_rgbd = seqbench.rgbd_workload(speed=1.0, n_base=args.bases, stagger=24)
    wl_st = seqbench.stereo_workload()
    head, second = (wl_st, wl_rgbd) if stereo_head else (wl_rgbd, wl_st)
    # 8192 sequences per GPU: 21.7 k frames/s against 19.3 k with 4096 and 22.9 k with 16384 (same code; local-BA calls of ~82 windows instead of ~41; 55 GB of host
    # memory for the maps, 53 s of pre-roll)
    S = args.seqs or (2048 if stereo_head else 8192)
    # A rank of the default RGB-D job holds ~70 GB of host memory (8192 per-sequence maps of ~6.3 MB + the inputs: DESIGN.md section 9).  On a node whose memory
    # does not hold that for every local rank the job would be killed, not slowed down: the per-rank sequence count is halved until it fits (reported in `config`).
    seqs_reduced_from = None
    if not args.seqs and not stereo_head and world > 1:
        try:
            with open("/proc/meminfo") as fh:
                host_mem_gb = next(int(ln.split()[1]) for ln in fh if ln.startswith("MemTotal")) / 1048576.0
            per_rank = 0.85 * host_mem_gb / max(1, local_world)
            while S > 1024 and 8.5e-3 * S + 8.0 > per_rank:     # (8.5 MB per sequence incl. allocator overhead and scratch, 8 GB of inputs / runtime)
                seqs_reduced_from = seqs_reduced_from or S
                S //= 2
        except Exception:
            pass
    G = args.handles or 8
    extras_on = rank == 0 and world == 1 and not args.no_extras
    host_phase = extras_on and not stereo_head            # the same warmed sequences continued with host-resident inputs
    post_frames = (2 + args.steps) if host_phase else 0
    n_frames = preroll + args.warmup + args.steps + post_frames
    log = (lambda m: (sys.stderr.write("[bench] " + m + "\n"), sys.stderr.flush())) if rank == 0 else None
    # ---- render the input streams on the host cores BEFORE the process touches the GPU (worker processes are forked) ----
    t_gen = time.perf_counter()
    # OSLAM_BENCH_SHARED_BASES=1 (multi-rank runs): the node's ranks replay ONE set of rendered base streams mapped from a shared-memory segment instead of
    # rendering a set each (seqbench.shared_base_sequences; every rank then runs rank 0's streams: identical work per rank).  Any failure falls back to private sets.
    seq_head = None
    if world > 1 and os.environ.get("OSLAM_BENCH_SHARED_BASES"):
        try:
            seq_head = seqbench.shared_base_sequences(head, local_rank, local_world, S, n_frames, tag=os.environ.get("MASTER_PORT", "0"), workers=share, log=log)
        except Exception as ex:     # (no segment within the time limit, no room in /dev/shm, ...)
            sys.stderr.write("[bench] rank %d: shared base streams unavailable (%r): rendering a private set\n" % (rank, ex))
            seq_head = None
    if seq_head is None:
        seq_head = seqbench.base_sequences(head, rank, S, n_frames, workers=share)
    seq_second = None
    pre2 = 0
    if extras_on:            # second figure: the other stream shape in ITS steady state (rank 0 of a single-rank run only)
        S2, G2 = (2048, 8) if second is wl_st else (1024, 4)   # (stereo, same box: 512 / 4 -> 13.1 k frames/s, 1024 / 4 -> 16.2 k, 2048 / 4 -> 17.0 k, 2048 / 8 -> 19.8 k; 48 GB of HBM)
        pre2 = 40 if second is wl_st else 200
        seq_second = seqbench.base_sequences(second, rank, S2, pre2 + args.warmup + args.steps, workers=share)
    # Timed range of a CPU baseline = the stream frames the GPU leg's sequences cover in ITS timed region: sequence offsets 0 .. stagger, steps preroll + warmup ..
    # preroll + warmup + steps, i.e. stream frames [preroll + warmup, preroll + warmup + steps + stagger) — the same part of the path with a map of the same age
    # (later frames of these periodic paths revisit mapped places and insert a third of the keyframes).  Every base stream holds n_frames + stagger frames.
    cpu_n = args.cpu_frames or (args.steps + head.stagger)
    cpu_n2 = args.steps + second.stagger
    # second headline figure: 32 base renderings (800 distinct input streams): the first 8 are the headline's, 24 more are rendered
    seq_b32 = wl_b32 = None
    child_legs = extras_on and args.extras_child_processes
    want_b32 = extras_on and not stereo_head and not args.no_bases32 and args.bases < 32 and S >= 32
    if want_b32:
        wl_b32 = seqbench.rgbd_workload(speed=1.0, n_base=32, stagger=24)
    if want_b32 and not child_legs:
        seq_b32 = seqbench.base_sequences(wl_b32, rank, S, preroll + args.warmup + args.steps + post_frames, workers=share,
                                          have=seq_head if (rank == 0 and head.stagger == wl_b32.stagger) else None)
    t_gen = time.perf_counter() - t_gen
    if log:
        log("inputs rendered in %.1f s" % t_gen)

    def _input_gb(seqs, seen):   # host bytes of a leg's rendered base streams (arrays shared with an earlier leg counted once)
        n = 0
        for q in (seqs or {}).values():
            for v in q.values():
                if hasattr(v, "nbytes") and id(v) not in seen:
                    seen.add(id(v)); n += v.nbytes
        return round(n / 2**30, 2)
    _seen = set()
    inputs_host_gb = {"headline": _input_gb(seq_head, _seen), "bases32_extra": _input_gb(seq_b32, _seen), "second_workload": _input_gb(seq_second, _seen)}

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = "nccl"        # = RCCL on ROCm
    if local_rank >= ndev:
        if not os.environ.get("OSLAM_BENCH_SHARE_GPU"):
            raise SystemExit("rank %d: only %d GPU(s) visible (set OSLAM_BENCH_SHARE_GPU=1 to rehearse several ranks on one card)" % (local_rank, ndev))
        local_rank %= ndev
    if world > ndev:
        backend = "gloo"    # rehearsal of several ranks on one card: RCCL refuses two ranks on the same device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    kt = {}
    extras_failed = []   # names of extra legs that raised (the line records their errors; --strict turns them into a non-zero exit)

    mem_gb = {}
    rss_gb = {}

    def run(wl, seqs, S_, G_, tag, preroll_=0, post_frames_=0, post_=None, drop_host_inputs=(), trim=False):
        snap = {}

        def stage_totals(systems):
            st, co = {}, {}
            for sy in systems:
                for k, v in sy.stage_seconds().items():
                    st[k] = st.get(k, 0.0) + v
                for k, v in sy.stage_seconds(cpu=True).items():
                    co[k] = co.get(k, 0.0) + v
            return st, co

        def reset_timers(systems):
            for sy in systems:
                sy.kernel_times(True)
            snap["t0"] = stage_totals(systems)
        threads = int(os.environ.get("OSLAM_BENCH_HOST_THREADS", "0")) or max(1, share // G_)
        summ, rec, systems, extra = seqbench.run_rank(wl, lambda cfg: slam.System(cfg), rank, world, S_, G_, args.steps, args.warmup, True, device,
                                                      host_threads=threads, sequences=seqs, after_warmup=reset_timers, coll_on_device=backend == "nccl",
                                                      preroll=preroll_, post_frames=post_frames_, post=post_, progress=log,
                                                      local_mapping=slam.LM_DEFERRED if args.lm == "deferred" else slam.LM_SYNC, drop_host_inputs=drop_host_inputs)
        tot = extra.get("post", {}).get("kernel_times_timed") if isinstance(extra.get("post"), dict) else None
        if tot is None:
            tot = {}
            for sy in systems:
                for g, v in sy.kernel_times(False).items():
                    a = tot.setdefault(g, dict(ms=0.0, launches=0.0, work=0.0))
                    for k in v:
                        a[k] += v[k]
        kt[tag] = tot
        stages, cores_s = stage_totals(systems)
        summ["stage_seconds_sum_over_handles"] = {k: round(v, 4) for k, v in stages.items()}
        summ["stage_core_seconds_sum_over_handles"] = {k: round(v, 4) for k, v in cores_s.items()}
        # the same sums over the TIMED steps only (the totals above include the pre-roll, the warm-up and any later leg)
        t1 = extra.get("post", {}).get("stages_timed_end") if isinstance(extra.get("post"), dict) else None
        t1 = t1 or (stages, cores_s)
        if "t0" in snap:
            summ["stage_seconds_timed_sum_over_handles"] = {k: round(t1[0][k] - snap["t0"][0].get(k, 0.0), 4) for k in t1[0]}
            summ["stage_core_seconds_timed_sum_over_handles"] = {k: round(t1[1][k] - snap["t0"][1].get(k, 0.0), 4) for k in t1[1]}
        reuse = [sy.local_map_reuse() for sy in systems]
        summ["local_map_reuse_frac"] = round(sum(r[0] for r in reuse) / max(1, sum(r[1] for r in reuse)), 4)
        summ["host_threads_per_handle"] = threads
        summ["post"] = extra.get("post")
        try:   # device memory in use while the leg's systems are alive (maps, resident records, extractor buffers, local-BA arenas, the input images)
            free_b, total_b = torch.cuda.mem_get_info(device)
            mem_gb[tag] = round((total_b - free_b) / 2**30, 1)
        except Exception:
            mem_gb[tag] = None
        try:   # host memory of the process while the leg's systems are alive (their maps, the pinned staging blocks, every leg's rendered inputs)
            with open("/proc/self/status") as fh:
                vm = {ln.split(":")[0]: int(ln.split()[1]) for ln in fh if ln.startswith(("VmRSS", "VmHWM"))}
            rss_gb[tag] = {"rss": round(vm["VmRSS"] / 1048576.0, 2), "peak_so_far": round(vm["VmHWM"] / 1048576.0, 2)}
        except Exception:
            rss_gb[tag] = None
        for sy in systems:
            sy.close()
        del systems, extra
        import gc
        gc.collect()
        torch.cuda.empty_cache()   # (the leg's systems and input tensors are gone before the next leg allocates: left referenced, they made every later leg 1.2-2x slower)
        # malloc_trim(0) — handing the closed leg's tens of GB of small blocks back to the system — only where the caller asks for it (`trim`): it costs the NEXT leg some
        # speed (page faults while its maps grow: measured 46.6 -> 43.8 k for the 32-base leg) and is what keeps the peak of the whole run away from the box's memory
        # limit: without it the 32-base leg's maps come on top of the arenas the headline's maps left behind (105-107 GB; one run of the round was killed there).
        if trim:
            try:
                import ctypes
                ctypes.CDLL("libc.so.6").malloc_trim(0)
            except Exception:
                pass
        return summ, rec

    def host_inputs_phase(ctx):
        """The warmed sequences continued for 2 + K more steps with the inputs in pinned HOST memory (gray u8, raw 16-bit depth, one-bit-per-pixel masks):
        the upload is inside the timed region (the kernels read the pinned buffers over PCIe where they are)."""
        systems = ctx["systems"]
        ktm = {}
        for sy in systems:                      # the headline's kernel groups end here
            for g, v in sy.kernel_times(True).items():
                a = ktm.setdefault(g, dict(ms=0.0, launches=0.0, work=0.0))
                for k in v:
                    a[k] += v[k]
        st_end, co_end = {}, {}
        for sy in systems:
            for k, v in sy.stage_seconds().items():
                st_end[k] = st_end.get(k, 0.0) + v
            for k, v in sy.stage_seconds(cpu=True).items():
                co_end[k] = co_end.get(k, 0.0) + v
        out = {"kernel_times_timed": ktm, "stages_timed_end": (st_end, co_end)}
        try:
            hc, nbytes = seqbench.host_input_calls(ctx, n_frames - post_frames, post_frames)
            t0 = n_frames - post_frames
            ctx["phase"](t0, t0 + 2, hc)
            ctx["sync"]()
            w0 = ctx["window_totals"]()
            ts = time.perf_counter()
            ctx["phase"](t0 + 2, t0 + post_frames, hc)
            ctx["sync"]()
            dt = time.perf_counter() - ts
            fr = S * args.steps
            out["host_inputs"] = {"frames_per_s": round(fr / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 3), "steps": args.steps,
                                  "bytes_per_frame": int(nbytes), "pcie_GBs": round(nbytes * fr / dt / 1e9, 2),
                                  "lba_windows_timed": seqbench.window_stats(ctx["window_totals"](), w0),
                                  "inputs": "pinned host memory: 8-bit gray, raw 16-bit depth (DepthMapFactor 5000, src/Tracking.cc:262), 3 instance masks as one bit per "
                                            "pixel; read by the Frame::Frame / object kernels over PCIe inside the timed region, the same warmed sequences as the headline"}
        except Exception as ex:      # the extra leg must not break the headline line
            out["host_inputs"] = {"error": repr(ex)}; extras_failed.append("host_inputs")
        return out

    summ, rec = run(head, seq_head, S, G, "head", preroll, post_frames, host_inputs_phase if host_phase else None)
    second_out = cold = bases32 = None
    if extras_on:
        if log:
            log("headline done: %.1f frames/s" % summ["frames_per_s"])
        if preroll > 0 and args.cold:     # the cold-start regime of rounds 1-2 on the same streams (maps at most warmup + steps frames old)
            c, _ = run(head, seq_head, S, G, "cold")
            cold = {"frames_per_s": round(c["frames_per_s"], 1), "ms_per_step": round(c["ms_per_step"], 3), "lba_windows_timed": c["lba_windows_timed"],
                    "note": "steps %d..%d of empty maps (no pre-roll): the regime bench.py timed in rounds 1-2 (there at twice the motion per frame)" % (args.warmup, args.warmup + args.steps)}
        # --extras-child-processes: the second workload and the 32-base leg each in a CHILD PROCESS (this program again, headline only).  Built while the third
        # 8192-sequence leg of one process ran 1.2-2x slower than alone (cause: the previous legs' device memory was still referenced, see run()); measured worse: with
        # the parent process alive on the card the stereo child ran at 4.5 k and the 32-base child at 33.8 k frames/s.  Default: in-process.
        def child_leg(tag, extra):
            import subprocess
            try:
                torch.cuda.empty_cache()
            except Exception:
                pass
            cmd = [sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--steps", str(args.steps), "--warmup", str(args.warmup), "--lm", args.lm] + extra
            if log:
                log("%s leg in a child process ..." % tag)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError("child leg %s failed (rc %d): %s" % (tag, r.returncode, r.stderr[-1500:]))
            return json.loads(lines[-1])

        s2 = roof2 = None
        second_err = None
        if child_legs:
            try:
                c2 = child_leg("second workload", ["--workload", "stereo" if second is wl_st else "rgbd", "--seqs", str(S2), "--handles", str(G2)])
                s2 = {"frames_per_s": c2["value"], "ms_per_step": c2["ms_per_step"], "ate_rmse_m": c2["ate_rmse_m"], "keyframes": c2["keyframes"], "local_bas": c2["local_bas"],
                      "lost_frames": c2["lost_frames"], "lba_windows_timed": c2["lba_windows_timed"], "stage_seconds_timed_sum_over_handles": c2.get("stage_seconds_timed_sum_over_handles")}
                roof2 = c2["roofline"]
            except Exception as ex:
                extras_failed.append("second_workload"); s2 = None
                second_err = repr(ex)
            if want_b32:
                try:
                    cb = child_leg("32-base", ["--bases", "32", "--seqs", str(S), "--handles", str(G)])
                    bases32 = {"frames_per_s": cb["value"], "ms_per_step": cb["ms_per_step"], "distinct_streams_per_gpu": min(S, wl_b32.n_base * (wl_b32.stagger + 1)),
                               "replicas_per_stream": round(S / min(S, wl_b32.n_base * (wl_b32.stagger + 1)), 1), "lba_windows_timed": cb["lba_windows_timed"],
                               "keyframes": cb["keyframes"], "local_bas": cb["local_bas"], "lost_frames": cb["lost_frames"], "ate_rmse_m": cb["ate_rmse_m"],
                               "device_ms_by_group": {g: round(v["device_ms"], 1) for g, v in cb["roofline"]["groups"].items()}, "process": "child",
                               "note": "the headline configuration with 32 base renderings instead of 8 (its own process): ~10 sequences per distinct input frame instead of ~41 (less "
                                       "cache sharing of the input images in Frame::Frame); the 24 other scenes insert fewer keyframes, so compare `lba_windows_timed` before comparing frames/s"}
                except Exception as ex:
                    bases32 = {"error": repr(ex)}; extras_failed.append("bases32")
        else:
            # (the second workload runs right behind the headline, before the 32-base leg: after TWO 8192-sequence legs have come and gone in the process its device-side
            # stages ran 1.5-2x slower — 12.5 / 16.1 k against 22.8 k frames/s for the same workload in a fresh process: gpurun_out of round 5)
            s2, _ = run(second, seq_second, S2, G2, "second", pre2, trim=seq_b32 is not None)
            roof2 = roofline_of(kt["second"], s2, second is wl_st)
            if seq_b32 is not None:
                try:
                    sb, _ = run(wl_b32, seq_b32, S, G, "bases32", preroll, drop_host_inputs=[b for b in seq_b32 if b not in seq_head], trim=True)
                    kb = kt["bases32"]
                    bases32 = {"frames_per_s": round(sb["frames_per_s"], 1), "ms_per_step": round(sb["ms_per_step"], 3), "distinct_streams_per_gpu": min(S, wl_b32.n_base * (wl_b32.stagger + 1)),
                               "replicas_per_stream": round(S / min(S, wl_b32.n_base * (wl_b32.stagger + 1)), 1), "lba_windows_timed": sb["lba_windows_timed"],
                               "keyframes": sb["keyframes"], "local_bas": sb["local_bas"], "lost_frames": sb["lost_frames"], "ate_rmse_m": round(sb["ate_rmse_m"], 6),
                               "device_ms_by_group": {g: round(v["ms"], 1) for g, v in kb.items()},
                               "note": "the headline configuration with 32 base renderings instead of 8: ~10 sequences per distinct input frame instead of ~41 (less cache sharing of the "
                                       "input images in Frame::Frame); the 24 other scenes insert fewer keyframes, so compare `lba_windows_timed` before comparing frames/s"}
                except Exception as ex:
                    bases32 = {"error": repr(ex)}; extras_failed.append("bases32")
                seq_b32 = None
        cpu2 = None
        if not args.no_cpu_baseline:
            if log:
                log("CPU baseline of the second workload ...")
            cpu2 = cpu_baseline(second, seq_second, pre2 + args.warmup, cpu_n2, slam.LM_DEFERRED if args.lm == "deferred" else slam.LM_SYNC, two_thread_streams=2)
        if s2 is None:
            second_out = {"error": second_err}
        else:
          second_out = {"workload": "%s, %d sequences per GPU in %d handles, %d features, local BA on every keyframe; steady state: %d untimed steps, then steps %d..%d timed; "
                                  "BASELINE.json configs[%s]" % (second.name, S2, G2, second.nFeatures, pre2, pre2 + args.warmup, pre2 + args.warmup + args.steps,
                                                                 "3]/[4" if second is wl_st else "2"),
                      "regime": "steady_state", "preroll_steps": pre2,
                      "frames_per_s": round(s2["frames_per_s"], 1), "ms_per_step": round(s2["ms_per_step"], 3), "ate_rmse_m": round(s2["ate_rmse_m"], 6),
                      "keyframes": s2["keyframes"], "local_bas": s2["local_bas"], "lost_frames": s2["lost_frames"], "lba_windows_timed": s2["lba_windows_timed"],
                      "roofline": {k: roof2[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_us", "algorithmic_work_per_launch", "work_unit", "groups")},
                      "cpu_baseline_stereo" if second is wl_st else "cpu_baseline_rgbd": cpu2,
                      "stage_seconds_timed_sum_over_handles": s2.get("stage_seconds_timed_sum_over_handles")}

    out = None
    if rank == 0:
        roof = roofline_of(kt["head"], summ, head is wl_st)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            if log:
                log("CPU baseline ...")
            cpu = cpu_baseline(head, seq_head, preroll + args.warmup, cpu_n, slam.LM_DEFERRED if args.lm == "deferred" else slam.LM_SYNC)
            if extras_on:
                try:
                    cpu["single_sequence_hip"] = single_sequence(head, seq_head[0], preroll + args.warmup, cpu_n, local_rank)
                except Exception as ex:
                    cpu["single_sequence_hip"] = {"error": repr(ex)}; extras_failed.append("single_sequence_hip")
        front = None
        if extras_on and head is wl_rgbd:
            try:
                q = seq_head[0]
                front = frontend_stage(q["gray"][:40], q["Twc"][:40], q["depth"][:40], local_rank, args.steps)
            except Exception as ex:      # the stage entry must not break the headline line
                front = {"error": repr(ex)}; extras_failed.append("frontend")
        if extras_on:
            try:
                roof["lba_alone"] = lba_alone(local_rank)
                # the same operator at the batch sizes the deferred schedule's service runs it at (~120-160 windows per call) and at one window per CU
                roof["lba_alone_by_batch"] = [lba_alone(local_rank, 160, 2, 1), lba_alone(local_rank, 256, 2, 1), lba_alone(local_rank, 256, 2, 2)]
            except Exception as ex:      # (must not break the headline line)
                roof["lba_alone"] = roof.get("lba_alone") or {"error": repr(ex)}
                roof["lba_alone_by_batch"] = {"error": repr(ex)}; extras_failed.append("lba_alone")
            try:                         # the card's measured fp64 issue rates beside the datasheet figure `peak` is taken from
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import mfma_f64_rate
                roof["fp64_peak_measured"] = mfma_f64_rate.measure(local_rank)
            except Exception as ex:
                roof["fp64_peak_measured"] = {"error": repr(ex)}; extras_failed.append("fp64_peak_measured")
        regime = ("steady state: every sequence is advanced %d untimed steps before the warm-up, so the timed steps are frames %d..%d of every sequence (SURVEY.md §8(d): "
                  "frame >= 200 of the S1 stream at <= 2 cm / 0.5 deg per frame)" % (preroll, preroll + args.warmup, preroll + args.warmup + args.steps)) if preroll > 0 else \
                 ("cold start: steps %d..%d of empty maps" % (args.warmup, args.warmup + args.steps))
        out = {"metric": "frames/sec tracking+localBA", "value": round(summ["frames_per_s"], 1), "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(summ["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s stream through the oslam_slam driver (Tracking::Track + LocalMapping::Run incl. LocalBundleAdjustment on every keyframe): "
                                      "%d sequences per GPU in %d handles (one host thread + %d workers each), %d ORB features, images resident in HBM; %s; "
                                      "BASELINE.json configs[%s]" % (head.name, S, G, summ["host_threads_per_handle"] - 1, head.nFeatures, regime, "2" if head is wl_rgbd else "3]/[4"),
                          "regime": "steady_state" if preroll > 0 else "cold_start", "preroll_steps": preroll, "preroll_s": round(summ["preroll_s"], 1),
                          "sequences_per_gpu": S, "sequences_per_gpu_reduced_from_for_host_memory": seqs_reduced_from, "frames_per_step": S * world, "local_mapping_schedule": args.lm,
                          "distinct_streams_per_gpu": min(S, head.n_base * (head.stagger + 1)), "replicas_per_stream": round(S / min(S, head.n_base * (head.stagger + 1)), 1),
                          "inputs_note": "the %d sequences of a GPU replay %d base renderings at %d frame offsets: every distinct input frame of a step is read by ~%.0f sequences "
                                         "(maps, keyframes and local-BA windows are per sequence and not shared); --bases N renders more base streams for an A/B of the cache "
                                         "effect on the Frame::Frame group" % (S, min(head.n_base, S), head.stagger + 1, S / min(S, head.n_base * (head.stagger + 1))), "host_cores": cores, "host_cpus_of_rank": ncpu,
                          "arithmetic": "u8/int front-end, fp64 optimisers (dtype names the optimisers' type)",
                          "parallelism": "independent sequences sharded over ranks (sequence i -> rank i mod N); no data-path collective"},
               "lba_windows_timed": summ["lba_windows_timed"],
               "ate_rmse_m": round(summ["ate_rmse_m"], 6), "keyframes": summ["keyframes"], "local_bas": summ["local_bas"], "lost_frames": summ["lost_frames"],
               "map_violations": summ["map_violations"], "semantic_edges": summ["semantic_edges"],
               "stage_seconds_sum_over_handles": summ["stage_seconds_sum_over_handles"],
               "stage_core_seconds_sum_over_handles": summ["stage_core_seconds_sum_over_handles"],
               "stage_seconds_timed_sum_over_handles": summ.get("stage_seconds_timed_sum_over_handles"),
               "stage_core_seconds_timed_sum_over_handles": summ.get("stage_core_seconds_timed_sum_over_handles"),
               "local_map_reuse_frac": summ["local_map_reuse_frac"],
               "per_rank": [{"rank": int(r[0]), "frames": int(r[2]), "elapsed_s": round(float(r[3]), 4), "local_bas": int(r[5]), "ate_rmse_m": round(float(r[7]), 6)} for r in rec],
               "roofline": roof, "cpu_baseline": cpu,
               "host_inputs": (summ.get("post") or {}).get("host_inputs") if isinstance(summ.get("post"), dict) else None,
               "cold_start": cold, "bases32": bases32, "stereo" if second is wl_st else "rgbd": second_out, "frontend": front, "extras_failed": extras_failed,
               "input_render_s": round(t_gen, 1), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "device_mem_used_gb_after_headline": mem_gb.get("head"), "host_max_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1048576.0, 2),
               # (the line above is the peak over ALL legs of this process — each leg's maps are freed into the allocator's arenas, not back to the system — ; the headline
               # job's own footprint, inputs of the other legs included, is the figure below, taken while its 8192 maps are alive)
               "host_rss_gb_headline_leg": rss_gb.get("head"),
               # every leg's inputs are rendered (forked workers) BEFORE the process touches the GPU, so the figure above contains the other legs' inputs too:
               "host_input_arrays_gb": inputs_host_gb}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if args.strict and extras_failed:
        raise SystemExit("extra legs failed: " + ", ".join(extras_failed))


def _lba_traffic():
    """Mean memory-side bytes per launch of an LM trial of the local-BA kernels (committed PMC summary; None when it is missing)."""
    try:
        return int(json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_lba_traffic.json")))["mean_bytes_per_launch_of_a_trial"])
    except Exception:
        return None


def _mfma_counters():
    """MFMA counters of the reduced-system solver the timed windows go through (k_w_chol_lds_mfma: LDS-resident system, v_mfma_f64_16x16x4_f64 row panels and
    trailing updates) from the committed rocprofv3 --pmc pass over one call of 40 steady-state-shaped windows (tools/lba_win_prof.py MODES=1 NB=40: n = 156;
    SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES; profiles/r05_pmc_lba_mfma.json, tools/gpu/final_prof.sh)."""
    path = os.path.join(ROOT, "profiles", "r05_pmc_lba_mfma.json")
    try:
        e = json.load(open(path))["k_w_chol_lds_mfma"]
        return {"kernel": "k_w_chol_lds_mfma (v_mfma_f64_16x16x4_f64 row panels + trailing updates of the LDS-resident reduced camera system, n = 156: the timed path's solver)",
                "mfma_util": round(e["mfma_util_of_occupied_cus"], 4), "mfma_util_chip": round(e["mfma_util_chip"], 5),
                "mfma_busy_cycles_per_launch": int(e["SQ_VALU_MFMA_BUSY_CYCLES"]), "mfma_mops_f64_per_launch": int(e["SQ_INSTS_VALU_MFMA_MOPS_F64"]),
                "mean_launch_us": round(e["mean_us"], 1),
                "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 4 SIMDs x CUs the launch occupies); counters, not pencil arithmetic: one workgroup "
                              "per window, bound by the per-panel critical path (16x16 diagonal factor + inverse on one wavefront: chol16_aug), not by the matrix pipe",
                "source": "profiles/r05_pmc_lba_mfma.json"}
    except Exception:
        return None


def roofline_of(k, summ, stereo):
    """Roofline of the dominant kernel group of the timed region (HIP events on the launch streams, rank 0)."""
    names = {"frames": ("hbm", "Frame::Frame (k_resize_lds, k_fast_cells_wave, k_blur_strip, k_octree, k_orient_describe, undistort, depth lookup%s)"
                        % (", stereo association" if stereo else "")),
             "pose_opt": ("fp64-valu", "k_pose_optimize"),
             "lba": ("fp64-valu", "local BA (lba.hip)")}
    dom = max(names, key=lambda g: k[g]["ms"])
    bound, kname = names[dom]
    ms, launches, work = k[dom]["ms"], max(k[dom]["launches"], 1.0), k[dom]["work"]
    if bound == "hbm":
        achieved, peak, unit = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = work / (ms * 1e-3) / 1e12, FP64_PEAK_TFLOPS, "TFLOP/s"
    group_tab = {}
    for g, v in k.items():
        e = {"device_ms": round(v["ms"], 3), "launches": int(v["launches"])}
        if g == "frames" and v["ms"] > 0:
            e["GBs"] = round(v["work"] / (v["ms"] * 1e-3) / 1e9, 1)
        if g in ("pose_opt", "lba") and v["ms"] > 0:
            e["fp64_TFLOPs"] = round(v["work"] / (v["ms"] * 1e-3) / 1e12, 4)
            e["flop"] = int(v["work"])
        group_tab[g] = e
    stream_ratio = sum(v["ms"] for v in k.values()) / (summ["elapsed_s"] * 1e3)
    total_ms = sum(v["ms"] for v in k.values())
    for g in group_tab:
        group_tab[g]["share"] = round(k[g]["ms"] / max(total_ms, 1e-9), 4)
    pipe = {"hbm": "hbm", "fp64-valu": "fp64 vector ALU (v_fma_f64 / v_mul_f64 / v_add_f64)"}[bound]
    if bound == "fp64-valu":
        bound = "mfma"   # the contract's two names are "hbm" | "mfma": `mfma` stands for the flop-bound case, `pipe` says which pipe executes the flops
    return {"bound": bound, "pipe": pipe, "peak_note": "HBM3E 8 TB/s" if bound == "hbm" else "fp64 vector ALU peak = fp64 matrix (MFMA) peak = 78.6 TFLOP/s on CDNA4 (datasheet; `fp64_peak_measured` = what "
            "tools/mfma_f64_rate.py reaches on this card); the linearisation / Schur / update kernels of this group issue v_fma_f64 / v_mul_f64 / v_add_f64, the reduced "
            "camera solve (k_w_chol_lds_mfma at 133..186 unknowns) v_mfma_f64_16x16x4_f64",
            "kernel": kname, "achieved": round(achieved, 4), "peak": peak, "unit": unit, "frac": round(achieved / peak, 5),
            "traffic": _pmc(kname.split(" ")[0].split(",")[0]) if bound == "hbm" else _lba_traffic(), "launch_us": round(ms / launches * 1e3, 1),
            "traffic_note": None if bound == "hbm" else "memory-side bytes per launch of the local-BA kernels (mean over the 6 launches of an LM trial) from the committed rocprofv3 "
            "--pmc FETCH_SIZE / WRITE_SIZE passes of ONE call of 40 steady-state-shaped windows (profiles/r05_pmc_lba_traffic.json, tools/pmc_lba_traffic.py); the "
            "calls of this run carry ~82 windows: scale by the windows per call",
            "algorithmic_work_per_launch": int(work / launches), "work_unit": "bytes" if bound == "hbm" else "fp64 flop",
            "groups": group_tab,
            # kernel time summed over the rank's streams (8 handles + the local-BA service run side by side) divided by the wall time of the timed region: how many
            # streams the card serves at once on average, NOT a busy fraction (it exceeds 1)
            "stream_time_over_wall_time": round(stream_ratio, 4), "mfma": _mfma_counters(),
            "note": "device ms summed over the rank's handles (their streams overlap); `lba` and `pose_opt` work = SURVEY.md §8(d) flop model x the LM "
                    "iterations / trials the kernels report"}


if __name__ == "__main__":
    main()
