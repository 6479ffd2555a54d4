#!/usr/bin/env python3
"""bench.py — frames/s of the MI355X-native front-end on BASELINE.json configs[1] (S2):
synthetic 640x480 stream, 1000 ORB features, ORBextractor + ORBmatcher (SearchByProjection
against the previous frame with the ground-truth pose), inputs resident in HBM.

A "step" is one pass of the hot path over one batch of B frames: batched extraction of all B
frames, then for every frame the last-frame projection + windowed Hamming search.
Contract: python bench.py --gpus N --steps K --warmup W ; one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS = 640, 480, 1000, 8
FX, FY, CX, CY, BF = 520.908620, 521.007327, 325.141442, 249.701764, 40.0   # reference Examples/RGB-D/TUM2.yaml:8-18
Z0 = 2.0
TH = 15.0          # reference src/Tracking.cc:962-966 (RGB-D: th=15)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def cpu_baseline(frames, offs, n_sample):
    """Oracle (CPU restatement, 1 thread) on a bounded sample of the same workload."""
    from oracle import oracle_py as O
    O.build()
    oe = O.OrbExtractor(NFEAT, 1.2, NLEVELS, 20, 7)
    sf = oe.tables()["scale"]
    cam = (FX, FY, CX, CY, BF, BF / FX)
    bounds = (0.0, 0.0, float(W), float(H))
    # warm
    oe.extract(frames[0])
    t0 = time.perf_counter()
    prev = None
    done = 0
    nB = len(frames)
    for j in range(n_sample):
        i = j % nB   # the stream is replayed if the sample is longer than the batch
        k, d = oe.extract(frames[i])
        if prev is not None:
            kl, dl, ol = prev
            du, dv = (offs[i] - ol).astype(np.float64)
            Xw = np.stack([(kl["x"] - CX) * Z0 / FX, (kl["y"] - CY) * Z0 / FY, np.full(len(kl), Z0)], 1).astype(np.float32)
            Tcw = np.eye(4, dtype=np.float32)
            Tcw[0, 3] = -du * Z0 / FX
            Tcw[1, 3] = -dv * Z0 / FY
            has = np.full(len(kl), 3, np.uint8)
            q = O.project_last_frame(Xw, has, kl, dl, Tcw, np.eye(4, dtype=np.float32), cam, bounds, sf, TH, False)
            uR = (k["x"] - BF / Z0).astype(np.float32)
            O.search_by_projection(k, uR, d, None, bounds, q, 0.9, False, True)
        prev = (k, d, offs[i])
        done += 1
    dt = time.perf_counter() - t0
    return done / dt, dt


def pmc_traffic(kernel_substr, frames_per_launch):
    """HBM-side bytes per launch of one kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_fetch_size.csv + r01_pmc_write_size.csv; FETCH_SIZE / WRITE_SIZE are in KB and, for the
    4-byte-per-lane loads these kernels issue, FETCH_SIZE matched the algorithmic byte count 1:1 on the
    calibration kernels (blur: 927 KB read vs 950 KB algorithmic), so no x2 correction is applied).
    The profile was taken at 256 frames per launch; scaled linearly to this run's batch."""
    import csv
    tot = 0.0
    for f in ("r01_pmc_fetch_size.csv", "r01_pmc_write_size.csv"):
        path = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(path):
            return None
        rows = [r for r in csv.DictReader(open(path)) if kernel_substr in r["Kernel_Name"]]
        if not rows:
            return None
        gmax = max(int(r["Grid_Size"]) for r in rows)   # the 256-frame launches of the bench, not batch-1 calls of the extras
        vals = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == gmax]
        tot += sum(vals) / len(vals)
    return tot * 1024.0 * frames_per_launch / 256.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (profiling runs: every launch of a kernel is then the S2 batch)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from object_slam_amd import ORBextractor, ORBmatcher, synth
    from object_slam_amd.matcher import MatchFrames, MatchLast

    B = args.batch
    # one independent synthetic stream per rank (batch-of-sequences: no data-path collective)
    frames, offs = synth.make_stream(B, W, H, seed=synth.SEED + rank)
    pitch = (W + 63) // 64 * 64
    d_img = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
    d_img[:, :, :W] = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream

    ex = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, W, H, max_batch=B, device=local_rank)
    cap = ex.cap
    mt = ORBmatcher(0.9, True, max_keypoints=cap, max_queries=cap, max_batch=B, device=local_rank)
    sf = ex.GetScaleFactors()
    cam = (FX, FY, CX, CY, BF, BF / FX)
    d_kp, d_desc, d_cnt, d_status = ex.results_device()

    # ---- build "the map" once from a first extraction: last frame = previous frame of the stream ----
    ex.extract_batch_device(d_img.data_ptr(), B, pitch, pitch * H, st)
    torch.cuda.synchronize()
    host = [ex.fetch(b) for b in range(B)]
    counts = np.array([len(k) for k, _ in host], np.int32)
    last_keys = np.zeros((B, cap), dtype=host[0][0].dtype)
    last_desc = np.zeros((B, cap, 32), np.uint8)
    last_Xw = np.zeros((B, cap, 3), np.float32)
    last_has = np.zeros((B, cap), np.uint8)
    last_n = np.zeros(B, np.int32)
    Tcw = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    Tlw = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    uR = np.full((B, cap), -1, np.float32)
    for b in range(B):
        p = (b - 1) % B
        kl, dl = host[p]
        n = len(kl)
        last_keys[b, :n] = kl
        last_desc[b, :n] = dl
        last_Xw[b, :n, 0] = (kl["x"] - CX) * Z0 / FX
        last_Xw[b, :n, 1] = (kl["y"] - CY) * Z0 / FY
        last_Xw[b, :n, 2] = Z0
        last_has[b, :n] = 3
        last_n[b] = n
        du, dv = (offs[b] - offs[p]).astype(np.float64)
        Tcw[b, 0, 3] = -du * Z0 / FX
        Tcw[b, 1, 3] = -dv * Z0 / FY
        kc = host[b][0]
        uR[b, :len(kc)] = kc["x"] - BF / Z0
    t_keys = torch.from_numpy(last_keys.view(np.uint8).reshape(B, -1)).cuda()
    t_desc = torch.from_numpy(last_desc).cuda()
    t_Xw = torch.from_numpy(last_Xw).cuda()
    t_has = torch.from_numpy(last_has).cuda()
    t_n = torch.from_numpy(last_n).cuda()
    t_Tcw = torch.from_numpy(Tcw).cuda()
    t_Tlw = torch.from_numpy(Tlw).cuda()
    t_uR = torch.from_numpy(uR).cuda()

    fr = MatchFrames()
    fr.keysUn, fr.kp_stride, fr.uRight, fr.desc, fr.blocked = d_kp, cap, t_uR.data_ptr(), d_desc, None
    fr.n_kps, fr.n_kps_const = d_cnt, 0
    fr.minX, fr.minY, fr.maxX, fr.maxY = 0.0, 0.0, float(W), float(H)
    la = MatchLast()
    la.Xw, la.has_mp, la.keys, la.mp_desc = t_Xw.data_ptr(), t_has.data_ptr(), t_keys.data_ptr(), t_desc.data_ptr()
    la.kp_stride, la.n_kps, la.n_kps_const = cap, t_n.data_ptr(), 0
    import ctypes as C
    q_nq = C.c_void_p()
    from object_slam_amd._lib import check
    check(mt.L.oslam_match_results_device(mt.h, None, None, None, None, None, C.byref(q_nq)))

    ev_m0, ev_m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    match_ms = []

    def step(timed_match=False):
        ex.extract_batch_device(d_img.data_ptr(), B, pitch, pitch * H, st)
        if timed_match:
            ev_m0.record()
        mt.project_last_batch_device(la, t_Tcw.data_ptr(), t_Tlw.data_ptr(), cam, fr, sf, TH, False, B, st)
        mt.search_batch_device(fr, None, cap, q_nq.value, 0, B, False, True, st)
        if timed_match:
            ev_m1.record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + sync on both sides ----
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    from object_slam_amd.parallel import aggregate_stats
    total_frames, elapsed = aggregate_stats(elapsed, B * args.steps, device="cuda")

    if os.environ.get("OSLAM_MATCH_DEBUG"):   # phase stamps of the matcher kernel (profiling builds)
        import ctypes as _C
        dbg = (_C.c_longlong * 8)()
        mt.L.oslam_match_debug_counters(mt.h, dbg, 1)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        mt.L.oslam_match_debug_counters(mt.h, dbg, 1)
        print("MATCH_PHASES_us_per_launch", [round(v * 0.01 / 10, 1) for v in dbg], file=sys.stderr)
    # sanity on results (outside the timed region): matches found, no arena overflow
    nm, qm, qd, km, iters = mt.fetch(B // 2, cap, int(last_n[B // 2]), cap, int(counts[B // 2]), st)
    k_chk, _ = ex.fetch(B // 2)   # raises on overflow status
    assert len(k_chk) == counts[B // 2] and nm > 100, (len(k_chk), nm)

    # ---- per-kernel timing (HIP events on the launch stream), separate untimed passes ----
    roof = None
    if rank == 0:
        ex.set_profiling(1)
        PK = max(3, min(args.steps, 10))
        for _ in range(PK):
            step(timed_match=True)
            torch.cuda.synchronize()
            match_ms.append(ev_m0.elapsed_time(ev_m1))
        ms, nb, ni = ex.get_profile()
        ex.set_profiling(0)
        n_kp = float(counts.mean())
        Ptot = sum(ex.level_size(l)[0] * ex.level_size(l)[1] for l in range(NLEVELS))
        p0 = W * H
        pl = ex.level_size(NLEVELS - 1)[0] * ex.level_size(NLEVELS - 1)[1]
        # algorithmic bytes per frame of each kernel group
        alg = {"pyramid(K1)": (Ptot - pl) + (Ptot - p0), "fast_cells(K2/K3)": Ptot, "blur(K6)": 2 * Ptot,
               "octree(K4)": 0, "orient_describe(K5/K7)": (749 + 512 + 60) * n_kp}
        names = list(alg.keys())
        per_frame_us = {n: m / ni * 1e3 for n, m in zip(names, ms)}
        per_frame_us["match(K8-K10)"] = float(np.mean(match_ms)) / B * 1e3
        M = N = n_kp
        alg["match(K8-K10)"] = 44 * M + 36 * N + 12288 + 8 * M
        # matching work model: descriptor pairs compared per second against the v_bcnt_u32_b32 issue peak
        # (8 popcounts per pair; 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s -> 4.9 T pairs/s if nothing else issued)
        pairs = C.c_int64(0)
        check(mt.L.oslam_match_hamming_pairs(mt.h, B, C.byref(pairs)))
        match_pairs_per_s = pairs.value / (float(np.mean(match_ms)) * 1e-3)
        match_model = {"hamming_pairs_per_frame": round(pairs.value / B, 1), "hamming_pairs_per_s": round(match_pairs_per_s, 1),
                       "frac_of_bcnt_issue_peak": round(match_pairs_per_s * 8 / (256 * 4 * 16 * 2.4e9), 5)}
        dom = max(per_frame_us.keys(), key=lambda n: per_frame_us[n])
        achieved = alg[dom] / (per_frame_us[dom] * 1e-6) / 1e9
        # kernels of each timed group (the HIP events bracket the group; rocprofv3 lists the kernels separately)
        knames = {"pyramid(K1)": ["k_resize_lds"], "fast_cells(K2/K3)": ["k_fast_cells_wave", "k_fast_cells_ovf"],
                  "blur(K6)": ["k_blur_strip<false>", "k_blur_strip<true>"], "octree(K4)": ["k_octree"],
                  "orient_describe(K5/K7)": ["k_orient_describe"], "match(K8-K10)": ["k_project_last", "k_search_window"]}[dom]
        parts = [pmc_traffic(k, B) for k in knames]
        traffic = None if any(p is None for p in parts[:1]) else sum(p for p in parts if p is not None)
        roof = {"bound": "hbm", "kernel": dom, "kernels": knames, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if traffic is None else int(traffic),
                "algorithmic_bytes_per_launch": int(alg[dom] * B),
                "launch_us": round(per_frame_us[dom] * B, 1),
                "per_frame_us": {k: round(v, 3) for k, v in per_frame_us.items()},
                "alg_bytes_per_frame": {k: int(v) for k, v in alg.items()},
                "whole_path_GBs": round(ex.algorithmic_bytes(int(n_kp)) * B * world * args.steps / elapsed / 1e9, 1),
                "matching": match_model}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # rank 0, N=1 only (bounded sample)
        ns = args.cpu_sample
        v, dt = cpu_baseline(frames, offs, ns)
        cpu = {"value": round(v, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the same stream: oracle extract + SearchByProjection(Cur,Last), %.1f s" % (ns, dt)}

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        # other rows of the hot path (SURVEY.md §8(d) S5 and the pose optimisers), reported beside the headline
        try:
            from object_slam_amd import LocalBundleAdjuster, PoseOptimizer
            extras = {}
            # the same S2 step software-pipelined over two handle sets on two streams (batch i + 1 is extracted while batch i is matched): what a
            # caller with a queue of batches gets.  Not the headline: the kernels of two batches then overlap, so the per-kernel durations that
            # `roofline` reports (one batch at a time, as in the rocprofv3 summary) would no longer describe the timed region.
            ex2 = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, W, H, max_batch=B, device=local_rank)
            mt2 = ORBmatcher(0.9, True, max_keypoints=cap, max_queries=cap, max_batch=B, device=local_rank)
            d_kp2, d_desc2, d_cnt2, _ = ex2.results_device()
            fr2 = MatchFrames()
            fr2.keysUn, fr2.kp_stride, fr2.uRight, fr2.desc, fr2.blocked = d_kp2, cap, t_uR.data_ptr(), d_desc2, None
            fr2.n_kps, fr2.n_kps_const = d_cnt2, 0
            fr2.minX, fr2.minY, fr2.maxX, fr2.maxY = 0.0, 0.0, float(W), float(H)
            q_nq2 = C.c_void_p()
            check(mt2.L.oslam_match_results_device(mt2.h, None, None, None, None, None, C.byref(q_nq2)))
            ps = [torch.cuda.Stream(), torch.cuda.Stream()]
            psets = [(ex, mt, fr, q_nq, ps[0].cuda_stream), (ex2, mt2, fr2, q_nq2, ps[1].cuda_stream)]

            def pstep(i):
                e_, m_, f_, q_, s_ = psets[i & 1]
                e_.extract_batch_device(d_img.data_ptr(), B, pitch, pitch * H, s_)
                m_.project_last_batch_device(la, t_Tcw.data_ptr(), t_Tlw.data_ptr(), cam, f_, sf, TH, False, B, s_)
                m_.search_batch_device(f_, None, cap, q_.value, 0, B, False, True, s_)

            for i in range(4):
                pstep(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                pstep(i)
            torch.cuda.synchronize()
            extras["pipelined_two_batches_frames_per_s"] = round(B * args.steps / (time.perf_counter() - t1), 1)
            k2, _ = ex2.fetch(B // 2)
            assert len(k2) == counts[B // 2]
            ex2.close(); mt2.close()
            PB, PN = 128, 1000
            probs = [synth.make_pose_problem(100 + i, N=PN) for i in range(PB)]
            tt = lambda k, dt: torch.from_numpy(np.stack([q[k] for q in probs]).astype(dt)).cuda()
            a_T, a_X, a_o, a_i, a_h = tt("Tcw", np.float32), tt("Xw", np.float32), tt("obs", np.float32), tt("invSigma2", np.float32), tt("has_mp", np.uint8)
            po = PoseOptimizer(max_points=PN, max_batch=PB, device=local_rank)
            for _ in range(2):
                po.optimize_batch_device(PB, PN, None, PN, a_T.data_ptr(), a_X.data_ptr(), a_o.data_ptr(), a_i.data_ptr(), a_h.data_ptr(), probs[0]["K"], st)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                po.optimize_batch_device(PB, PN, None, PN, a_T.data_ptr(), a_X.data_ptr(), a_o.data_ptr(), a_i.data_ptr(), a_h.data_ptr(), probs[0]["K"], st)
            torch.cuda.synchronize()
            extras["pose_optimization_frames_per_s"] = round(PB * 5 / (time.perf_counter() - t1), 1)
            extras["pose_optimization_config"] = "Optimizer::PoseOptimization, %d frames/launch x %d keypoints (80%% with map points, 10%% outliers)" % (PB, PN)
            q5 = synth.make_lba_problem(1234)
            ba = LocalBundleAdjuster(max_keyframes=64, max_points=8192, max_edges=65536, device=local_rank)
            lba_args = (q5["poses"], q5["fixed"], q5["points"], q5["edge_kf"], q5["edge_pt"], q5["edge_obs"], q5["edge_invSigma2"], q5["K"])
            ba.LocalBundleAdjustment(*lba_args)
            t1 = time.perf_counter()
            for _ in range(3):
                r5 = ba.LocalBundleAdjustment(*lba_args)
            extras["lba_S5_ms"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
            extras["lba_S5_config"] = "S5: 20 local + 20 fixed KF, 4000 points, %d edges, host-to-host incl. PCIe" % len(q5["edge_kf"])
            if not args.no_cpu_baseline:
                from oracle import oracle_py as O
                t1 = time.perf_counter()
                for q in probs[:10]:
                    O.pose_optimization(q["Tcw"], q["Xw"], q["obs"], q["invSigma2"], q["has_mp"], q["K"])
                extras["pose_optimization_cpu_oracle_frames_per_s"] = round(10 / (time.perf_counter() - t1), 1)
                t1 = time.perf_counter()
                O.local_bundle_adjustment(*lba_args)
                extras["lba_S5_cpu_oracle_ms"] = round((time.perf_counter() - t1) * 1e3, 1)
            # end-to-end tracking + local-BA harness (object_slam_amd/e2e.py), single sequence, host-driven
            from object_slam_amd import e2e
            ef, eo = synth.make_stream(64, W, H, seed=11)
            ecam = (FX, FY, CX, CY, BF)
            hb = e2e.HipBackend(W, H, device=local_rank)
            e2e.run_sequence(hb, ef[:4], eo[:4], ecam, Z0)
            trk, edt, eate = e2e.run_sequence(hb, ef, eo, ecam, Z0)
            extras["e2e_tracking_localBA_frames_per_s"] = round(len(ef) / edt, 1)
            extras["e2e_ate_rmse_m"] = round(eate, 6)
            extras["e2e_config"] = "S1-shaped synthetic RGB-D, 64 frames 640x480, 1 sequence, %d keyframes, %d local BAs (harness: e2e.py)" % (len(trk.kfs), trk.stats["lba_calls"])
            if not args.no_cpu_baseline:
                from oracle.oracle_backend import OracleBackend
                trc, cdt, cate = e2e.run_sequence(OracleBackend(W, H), ef[:24], eo[:24], ecam, Z0)
                extras["e2e_cpu_oracle_frames_per_s"] = round(24 / cdt, 2)
                extras["e2e_cpu_oracle_ate_rmse_m"] = round(cate, 6)
            # batch-of-sequences tracking + local mapping (include/oslam_slam.h): S sequences in lockstep on this GPU, images resident in HBM
            import ctypes as C
            from object_slam_amd import slam
            import threading
            SB, NF, NBASE, NG = 512, 60, 8, 4   # 4 handles x 128 sequences (measured: 256 in 4 -> 14.5 k, 384 in 6 -> 17.1 k, 512 in 4 -> 17.9 k, 512 in 8 -> 16.2 k frames/s)
            SG = SB // NG
            base = [synth.make_stream(NF, W, H, seed=11 + s, margin=1200) for s in range(NBASE)]
            d_base = [torch.from_numpy(b[0]).cuda() for b in base]
            d_depth = torch.full((H, W), Z0, dtype=torch.float32, device="cuda")
            nthr = max(1, min(16, os.cpu_count() or 1) // NG)
            groups = [slam.System(slam.make_config(W, H, SG, device=local_rank, host_threads=nthr)) for _ in range(NG)]
            s_poses = []
            torch.cuda.synchronize()

            def _drive(g):   # one host thread per handle: the bookkeeping of one group overlaps the kernels of the others (each handle has its own stream)
                dptr = [d_depth.data_ptr()] * SG
                for t in range(NF):
                    Tb, _ = groups[g].TrackRGBD_device([d_base[(g * SG + s) % NBASE][t].data_ptr() for s in range(SG)], W, dptr, W, [t / 30.0] * SG)
                    if g == 0:
                        s_poses.append(Tb[0].copy())

            t1 = time.perf_counter()
            ths = [threading.Thread(target=_drive, args=(g,)) for g in range(NG)]
            for th_ in ths:
                th_.start()
            for th_ in ths:
                th_.join()
            sdt = time.perf_counter() - t1
            ssys = groups[0]
            sst = ssys.stats(0)
            _, sTwc = ssys.trajectory(0)
            off = (base[0][1] - base[0][1][0]).astype(np.float64)
            sgt = np.stack([off[:, 0] * Z0 / FX, off[:, 1] * Z0 / FY, np.zeros(len(off))], 1)
            extras["slam_batched_frames_per_s"] = round(SB * NF / sdt, 1)
            extras["slam_batched_ate_rmse_m"] = round(e2e.horn_align_ate(sTwc[:, :, 3], sgt[:len(sTwc)]), 6)
            extras["slam_batched_config"] = ("oslam_slam driver (Tracking::Track + LocalMapping::Run control flow), %d sequences x %d frames: %d handles of %d sequences in lockstep, "
                                             "one host thread + %d workers per handle, 640x480 RGB-D synthetic, images in HBM; seq 0: %d keyframes, %d local BAs, %d points fused, %d culled"
                                             % (SB, NF, NG, SG, nthr, sst["keyframes_created"], sst["local_bas"], sst["points_fused"], sst["points_culled"]))
            extras["slam_batched_stage_seconds_handle0"] = {k: round(v, 4) for k, v in ssys.stage_seconds().items()}
            if not args.no_cpu_baseline:
                from oracle import oracle_py as O
                ocfg = slam.make_config(W, H, 1)
                oops = slam.SlamOps()
                assert O.lib().oo_slam_make_ops(C.byref(ocfg), C.byref(oops)) == 0
                osys = slam.System(ocfg, oops)
                NO = 30
                depth_h = np.full((H, W), Z0, np.float32)
                t1 = time.perf_counter()
                o_poses = []
                for t in range(NO):
                    To, _ = osys.TrackRGBD([base[0][0][t]], [depth_h], [t / 30.0])
                    o_poses.append(To[0].copy())
                odt = time.perf_counter() - t1
                _, oTwc = osys.trajectory(0)
                extras["slam_cpu_oracle_frames_per_s"] = round(NO / odt, 2)
                extras["slam_cpu_oracle_ate_rmse_m"] = round(e2e.horn_align_ate(oTwc[:, :, 3], sgt[:len(oTwc)]), 6)
                extras["slam_cpu_oracle_config"] = "same driver over the CPU oracle's operator table, 1 sequence x %d frames, 1 core" % NO
                # same frames, same driver: the HIP trajectory of sequence 0 against the oracle's
                extras["slam_hip_vs_oracle_max_abs_pose_diff"] = float(np.abs(np.array(s_poses[:NO]) - np.array(o_poses)).max())
            # S3/S4 shape (BASELINE.json configs[3],[4]): KITTI-shaped rectified stereo, 1241x376, 2000 features, STEREO sensor, local BA on every keyframe
            KW_, KH_, KD_, SK, NK, NKB = 1241, 376, 32, 64, 40, 4
            kbase = [synth.make_stereo_stream(NK, KW_, KH_, seed=21 + s, margin=600, disparity=KD_) for s in range(NKB)]
            kpitch = (KW_ + 63) // 64 * 64
            d_kl = [torch.zeros((NK, KH_, kpitch), dtype=torch.uint8, device="cuda") for _ in range(NKB)]
            d_kr = [torch.zeros((NK, KH_, kpitch), dtype=torch.uint8, device="cuda") for _ in range(NKB)]
            for b_ in range(NKB):
                d_kl[b_][:, :, :KW_] = torch.from_numpy(kbase[b_][0]).cuda()
                d_kr[b_][:, :, :KW_] = torch.from_numpy(kbase[b_][1]).cuda()
            KG = 2
            SKG = SK // KG
            kthr = max(1, min(16, os.cpu_count() or 1) // KG)
            kgroups = [slam.System(slam.make_config(KW_, KH_, SKG, cam=slam.KITTI00, nFeatures=2000, sensor=slam.STEREO, device=local_rank, host_threads=kthr)) for _ in range(KG)]
            kcfg = kgroups[0].cfg
            torch.cuda.synchronize()

            def _drive_k(g):
                for t in range(NK):
                    kgroups[g].TrackStereo([d_kl[(g * SKG + s) % NKB][t].data_ptr() for s in range(SKG)], [d_kr[(g * SKG + s) % NKB][t].data_ptr() for s in range(SKG)],
                                           [t / 10.0] * SKG, on_device=True, stride=kpitch)

            t1 = time.perf_counter()
            ths = [threading.Thread(target=_drive_k, args=(g,)) for g in range(KG)]
            for th_ in ths:
                th_.start()
            for th_ in ths:
                th_.join()
            kdt = time.perf_counter() - t1
            ksys = kgroups[0]
            kst = ksys.stats(0)
            _, kTwc = ksys.trajectory(0)
            kz0 = kcfg.bf / KD_
            koff = (kbase[0][2] - kbase[0][2][0]).astype(np.float64)
            kgt = np.stack([koff[:, 0] * kz0 / kcfg.fx, koff[:, 1] * kz0 / kcfg.fy, np.zeros(len(koff))], 1)
            extras["slam_stereo_batched_frames_per_s"] = round(SK * NK / kdt, 1)
            extras["slam_stereo_batched_ate_rmse_m"] = round(e2e.horn_align_ate(kTwc[:, :, 3], kgt[:len(kTwc)]), 6)
            extras["slam_stereo_batched_config"] = ("S3/S4 shape: %d KITTI-shaped stereo sequences x %d frames (2 handles in lockstep), 1241x376, 2000 features, KITTI00-02.yaml calibration, "
                                                    "plane at %.2f m; seq 0: %d keyframes, %d local BAs, %d lost frames" % (SK, NK, kz0, kst["keyframes_created"], kst["local_bas"], kst["lost_frames"]))
            extras["slam_stereo_batched_stage_seconds_handle0"] = {k: round(v, 4) for k, v in ksys.stage_seconds().items()}
            if not args.no_cpu_baseline:
                kocfg = slam.make_config(KW_, KH_, 1, cam=slam.KITTI00, nFeatures=2000, sensor=slam.STEREO)
                koops = slam.SlamOps()
                assert O.lib().oo_slam_make_ops(C.byref(kocfg), C.byref(koops)) == 0
                kosys = slam.System(kocfg, koops)
                NKO = 12
                t1 = time.perf_counter()
                for t in range(NKO):
                    kosys.TrackStereo([kbase[0][0][t]], [kbase[0][1][t]], [t / 10.0])
                extras["slam_stereo_cpu_oracle_frames_per_s"] = round(NKO / (time.perf_counter() - t1), 2)
        except Exception as ex:   # never let the side measurements break the headline line
            extras = {"error": repr(ex)}

    if rank == 0:
        out = {"metric": "frames/sec tracking front-end (ORBextractor+ORBmatcher)", "value": round(total_frames / elapsed, 1),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "S2: synthetic 640x480 stream, 1000 ORB feats, ORBextractor + ORBmatcher "
                                      "(SearchByProjection vs previous frame, GT pose), BASELINE.json configs[1]",
                          "batch_frames_per_step": B, "frames_per_gpu_per_step": B, "keypoints_per_frame": float(counts.mean()),
                          "matches_frame_mid": int(nm), "claim_fixpoint_iterations": int(iters),
                          "parallelism": "independent sequences, one per GPU; no data-path collective"},
               "roofline": roof, "cpu_baseline": cpu, "extras": extras}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()   # ranks leave together (rank 0 runs the profiling passes alone)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
