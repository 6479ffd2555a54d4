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
def cpu_baseline(wl, seq, n_frames):
    import ctypes as C
    from object_slam_amd import slam
    from object_slam_amd.e2e import horn_align_ate
    from oracle import oracle_py as O
    O.build()
    cfg = slam.make_config(wl.width, wl.height, 1, cam=wl.cam, nFeatures=wl.nFeatures, sensor=wl.sensor)
    ops = slam.SlamOps()
    assert O.lib().oo_slam_make_ops(C.byref(cfg), C.byref(ops)) == 0
    sysm = slam.System(cfg, ops)
    per = []
    n = min(n_frames, len(seq["gray"]))
    t0 = time.perf_counter()
    for t in range(n):
        t1 = time.perf_counter()
        if wl.sensor == slam.STEREO:
            sysm.TrackStereo([seq["gray"][t]], [seq["right"][t]], [t / wl.fps])
        else:
            sysm.TrackRGBD([seq["gray"][t]], [seq["depth"][t]], [t / wl.fps])
        per.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    _, Twc = sysm.trajectory(0)
    T0inv = np.linalg.inv(seq["Twc"][0])
    gt = np.array([T0inv @ x for x in seq["Twc"][:len(Twc)]])
    st = sysm.stats(0)
    per = np.array(per)
    # the reference prints median and mean tracking time per frame (Examples/RGB-D/rgbd_tum.cc:126-134)
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of one sequence of the same workload through the same driver over the CPU oracle's operator table, %.1f s; tracking AND local "
                      "mapping run on ONE core one after the other (the reference overlaps LocalMapping on a second thread and, for stereo, extracts the two images "
                      "on two threads: <= 3 busy cores, so its wall time per frame lies between the median and the mean below)" % (n, dt),
            "mean_ms_per_frame": round(float(per.mean()) * 1e3, 2), "median_ms_per_frame": round(float(np.median(per)) * 1e3, 2),
            "ate_rmse_m": round(horn_align_ate(Twc[:, :, 3], gt[:, :3, 3]), 6), "keyframes": st["keyframes_created"], "local_bas": st["local_bas"]}


# --------------------------------------------------------------------------------------------------------------------------------
# S2 stage entry: ORBextractor + ORBmatcher only (BASELINE.json configs[1]), per-kernel HBM table
# --------------------------------------------------------------------------------------------------------------------------------
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
           # committed PMC passes of this stage (profiles/r02_pmc_traffic.json): HBM bytes per frame and the VALU issue share of the kernels' own duration
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
    """Per-launch PMC figure of `kernel` from the committed summary of this round (profiles/r02_pmc_traffic.json: separate rocprofv3 --pmc passes of
    the S2 stage at 512 frames per launch — FETCH_SIZE, WRITE_SIZE, SQ_* — corrected as MI355X_MICROARCH.md §HBM prescribes); None when the summary has
    no row for it.  HBM bytes for the fp64 solver kernels were not collected (their bound is not HBM)."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(kernel, {}).get(field)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("rgbd", "stereo"), default="rgbd")
    ap.add_argument("--seqs", type=int, default=0, help="sequences per GPU (default: 4096 RGB-D / 512 stereo; measured 512 -> 9 k, 1024 -> 12 k, 2048 -> 15 k, 4096 -> 16 k, "
                                                        "8192 -> 17-20 k RGB-D frames/s: the per-stage fixed costs of the lockstep driver are amortised over more sequences)")
    ap.add_argument("--handles", type=int, default=0, help="driver handles per GPU, one host thread each (default 8 / 4)")
    ap.add_argument("--cpu-frames", type=int, default=240, help="frames of one sequence through the CPU oracle table for cpu_baseline (~11 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from object_slam_amd import seqbench, slam
    cores = os.cpu_count() or 1
    share = max(1, cores // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    wl_rgbd, wl_st = seqbench.rgbd_workload(n_base=16, stagger=12), seqbench.stereo_workload()
    head, second = (wl_rgbd, wl_st) if args.workload == "rgbd" else (wl_st, wl_rgbd)
    S = args.seqs or (4096 if head is wl_rgbd else 512)
    G = args.handles or (8 if head is wl_rgbd else 4)
    extras_on = rank == 0 and world == 1 and not args.no_extras
    S2, G2 = (512, 4) if second is wl_st else (1024, 4)
    n_frames = args.warmup + args.steps
    # ---- render the input streams on the host cores BEFORE the process touches the GPU (worker processes are forked) ----
    t_gen = time.perf_counter()
    seq_head = seqbench.base_sequences(head, rank, S, n_frames, workers=min(share, 16))
    seq_second = None if args.no_extras else seqbench.base_sequences(second, rank, S2, n_frames, workers=min(share, 16))   # every rank runs the second figure too
    cpu_seq = None
    if extras_on and not args.no_cpu_baseline and args.cpu_frames > len(seq_head[0]["gray"]):
        cpu_seq = head.make_sequence(0, args.cpu_frames)
    t_gen = time.perf_counter() - t_gen

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

    threads_per_handle = max(1, min(16, share) // G)
    kt = {}

    def run(wl, seqs, S_, G_, tag):
        def reset_timers(systems):
            for sy in systems:
                sy.kernel_times(True)
        summ, rec, systems, extra = seqbench.run_rank(wl, lambda cfg: slam.System(cfg), rank, world, S_, G_, args.steps, args.warmup, True, device,
                                                      host_threads=int(os.environ.get("OSLAM_BENCH_HOST_THREADS", "0")) or max(1, min(16, share) // G_), sequences=seqs, after_warmup=reset_timers, coll_on_device=backend == "nccl")
        tot = {}
        for sy in systems:
            for g, v in sy.kernel_times(False).items():
                a = tot.setdefault(g, dict(ms=0.0, launches=0.0, work=0.0))
                for k in v:
                    a[k] += v[k]
        kt[tag] = tot
        stages = {}
        for sy in systems:
            for k, v in sy.stage_seconds().items():
                stages[k] = stages.get(k, 0.0) + v
        summ["stage_seconds_sum_over_handles"] = {k: round(v, 4) for k, v in stages.items()}
        cores_s = {}
        for sy in systems:
            for k, v in sy.stage_seconds(cpu=True).items():
                cores_s[k] = cores_s.get(k, 0.0) + v
        summ["stage_core_seconds_sum_over_handles"] = {k: round(v, 4) for k, v in cores_s.items()}
        reuse = [sy.local_map_reuse() for sy in systems]
        summ["local_map_reuse_frac"] = round(sum(r[0] for r in reuse) / max(1, sum(r[1] for r in reuse)), 4)
        for sy in systems:
            sy.close()
        return summ, rec

    summ, rec = run(head, seq_head, S, G, "head")
    second_out = None
    if seq_second is not None:
        s2, _ = run(second, seq_second, S2, G2, "second")
        second_out = {"workload": "%s, %d sequences per GPU in %d handles, %d features, local BA on every keyframe" % (second.name, S2, G2, second.nFeatures),
                      "frames_per_s": round(s2["frames_per_s"], 1), "ms_per_step": round(s2["ms_per_step"], 3), "ate_rmse_m": round(s2["ate_rmse_m"], 6),
                      "keyframes": s2["keyframes"], "local_bas": s2["local_bas"], "lost_frames": s2["lost_frames"],
                      "stage_seconds_sum_over_handles": s2["stage_seconds_sum_over_handles"]}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel group of the timed region (HIP events on the launch streams, rank 0) ----
        k = kt["head"]
        names = {"frames": ("hbm", "Frame::Frame (k_resize_lds, k_fast_cells_wave, k_blur_strip, k_octree, k_orient_describe, undistort, depth lookup%s)"
                            % (", stereo association" if head is wl_st else "")),
                 "pose_opt": ("mfma", "k_pose_optimize"),
                 "lba": ("mfma", "local BA, every LM trial of all windows as eight launches (k_w_lin, k_w_ctrlA, k_w_edgeW, k_w_schur, k_w_chol / k_w_chol_mfma, "
                                 "k_w_update, k_w_eval, k_w_ctrlB)")}
        dom = max(names, key=lambda g: k[g]["ms"])
        bound, kname = names[dom]
        ms, launches, work = k[dom]["ms"], max(k[dom]["launches"], 1.0), k[dom]["work"]
        if bound == "hbm":
            achieved, peak, unit = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = work / (ms * 1e-3) / 1e12, FP64_PEAK_TFLOPS, "TFLOP/s"
        group_tab = {}
        for g in ("frames", "pose_opt", "lba", "search"):
            v = k[g]
            e = {"device_ms": round(v["ms"], 3), "launches": int(v["launches"])}
            if g == "frames" and v["ms"] > 0:
                e["GBs"] = round(v["work"] / (v["ms"] * 1e-3) / 1e9, 1)
            if g in ("pose_opt", "lba") and v["ms"] > 0:
                e["fp64_TFLOPs"] = round(v["work"] / (v["ms"] * 1e-3) / 1e12, 4)
                e["flop"] = int(v["work"])
            group_tab[g] = e
        busy = sum(v["ms"] for v in k.values()) / (summ["elapsed_s"] * 1e3)
        roof = {"bound": bound, "kernel": kname, "achieved": round(achieved, 4), "peak": peak, "unit": unit, "frac": round(achieved / peak, 5),
                "traffic": _pmc(kname.split(" ")[0].split(",")[0]) if bound == "hbm" else None, "launch_us": round(ms / launches * 1e3, 1),
                "algorithmic_work_per_launch": int(work / launches), "work_unit": "bytes" if bound == "hbm" else "fp64 flop",
                "groups": group_tab, "device_busy_frac_of_timed_region": round(busy, 4),
                "note": "device ms summed over the rank's handles (their streams overlap); `lba` and `pose_opt` work = SURVEY.md §8(d) flop model x the LM "
                        "iterations / trials the kernels report"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(head, cpu_seq if cpu_seq is not None else seq_head[0], args.cpu_frames)
        front = None
        if extras_on and head is wl_rgbd:
            try:
                q = seq_head[0]
                front = frontend_stage(q["gray"], q["Twc"], q["depth"], local_rank, args.steps)
            except Exception as ex:      # the stage entry must not break the headline line
                front = {"error": repr(ex)}
        out = {"metric": "frames/sec tracking+localBA", "value": round(summ["frames_per_s"], 1), "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(summ["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s stream through the oslam_slam driver (Tracking::Track + LocalMapping::Run incl. LocalBundleAdjustment on every keyframe): "
                                      "%d sequences per GPU in %d handles (one host thread + %d workers each), %d ORB features, images resident in HBM; "
                                      "BASELINE.json configs[%s]" % (head.name, S, G, threads_per_handle, head.nFeatures, "2" if head is wl_rgbd else "3]/[4"),
                          "sequences_per_gpu": S, "frames_per_step": S * world, "host_cores": cores,
                          "arithmetic": "u8/int front-end, fp64 optimisers (dtype names the optimisers' type)",
                          "parallelism": "independent sequences sharded over ranks (sequence i -> rank i mod N); no data-path collective"},
               "ate_rmse_m": round(summ["ate_rmse_m"], 6), "keyframes": summ["keyframes"], "local_bas": summ["local_bas"], "lost_frames": summ["lost_frames"],
               "map_violations": summ["map_violations"], "semantic_edges": summ["semantic_edges"],
               "stage_seconds_sum_over_handles": summ["stage_seconds_sum_over_handles"],
               "stage_core_seconds_sum_over_handles": summ["stage_core_seconds_sum_over_handles"],
               "local_map_reuse_frac": summ["local_map_reuse_frac"],
               "per_rank": [{"rank": int(r[0]), "frames": int(r[2]), "elapsed_s": round(float(r[3]), 4), "local_bas": int(r[5]), "ate_rmse_m": round(float(r[7]), 6)} for r in rec],
               "roofline": roof, "cpu_baseline": cpu, "stereo" if second is wl_st else "rgbd": second_out, "frontend": front,
               "input_render_s": round(t_gen, 1)}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
