"""GPU: the tracking + local-mapping driver over the HIP operator table (the product) against the same driver over the CPU
oracle's table, on the same seeded RGB-D streams.  Matching and culling decisions are integer work and must agree exactly;
poses agree within the 1e-4 relative tolerance of the optimisers as long as the discrete decisions are the same."""
import numpy as np
import pytest

from object_slam_amd import slam
from slam_common import H, W, ate, make_streams, oracle_ops, run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lm", [slam.LM_SYNC, slam.LM_DEFERRED], ids=["sync", "deferred"])
def test_hip_driver_matches_oracle_driver(oracle, lm):
    """lm = deferred: the local BA of keyframe t is solved by the process-wide service while frame t+1 is tracked (include/oslam_slam.h head comment); the oracle
    table has no asynchronous form and solves it at collection time — same windows, same results."""
    n, S = 30, 3
    streams = make_streams(S, n)
    cfg = slam.make_config(W, H, S, local_mapping=lm)
    hip = slam.System(cfg)
    ph, sh = run(hip, streams, n)
    cfg_o = slam.make_config(W, H, S, local_mapping=lm)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run(ora, streams, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all()
    for s in range(S):
        a, b = hip.stats(s), ora.stats(s)
        assert a["map_violations"] == 0
        assert a == b, (s, a, b)     # same keyframes, points, fusions, cullings, inlier counts
        ah, Th = ate(hip, cfg, streams, s)
        ao, To = ate(ora, cfg_o, streams, s)
        assert ah < 0.01 and ao < 0.01
        assert np.abs(Th - To).max() < 2e-4, np.abs(Th - To).max()      # 1e-4 relative on a ~2 m scene
    d = np.abs(ph - po).max()
    assert d < 2e-4, d


def test_hip_driver_device_resident_inputs():
    import torch
    n, S = 8, 4
    streams = make_streams(S, n)
    cfg = slam.make_config(W, H, S)
    host = slam.System(cfg)
    ph, _ = run(host, streams, n)
    dev = slam.System(slam.make_config(W, H, S))
    d_depth = torch.full((H, W), 2.0, dtype=torch.float32, device="cuda")
    pd = []
    for t in range(n):
        imgs = [torch.from_numpy(streams[s][0][t]).cuda() for s in range(S)]
        torch.cuda.synchronize()
        T, st = dev.TrackRGBD_device([i.data_ptr() for i in imgs], W, [d_depth.data_ptr()] * S, W, [t / 30.0] * S)
        pd.append(T.copy())
    assert np.array_equal(np.array(pd), ph)


@pytest.mark.parametrize("lm", [slam.LM_SYNC, slam.LM_DEFERRED], ids=["sync", "deferred"])
def test_hip_stereo_driver_matches_oracle_driver(oracle, lm):
    from slam_common import ate_stereo, make_stereo_streams, run_stereo, stereo_config
    n, S = 16, 2
    streams = make_stereo_streams(S, n)
    cfg = stereo_config(S, local_mapping=lm)
    hip = slam.System(cfg)
    ph, sh = run_stereo(hip, streams, n)
    cfg_o = stereo_config(S, local_mapping=lm)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run_stereo(ora, streams, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all()
    for s in range(S):
        a, b = hip.stats(s), ora.stats(s)
        assert a == b, (s, a, b)
        assert a["map_violations"] == 0 and a["keyframes_created"] >= 2
        ah, _ = ate_stereo(hip, cfg, streams, s)
        assert ah < 0.05, ah
    # 1e-4 relative on a scene at 12 m
    assert np.abs(ph - po).max() < 2e-3, np.abs(ph - po).max()


@pytest.mark.parametrize("lm", [slam.LM_SYNC, slam.LM_DEFERRED], ids=["sync", "deferred"])
def test_two_handles_on_two_threads_match_sequential_runs(lm):
    """Several driver handles on one GPU, each advanced by its own host thread (the deployment bench.py measures): every handle owns its
    stream and buffers, so the concurrent runs must reproduce the sequential ones bit for bit."""
    import threading
    n, S = 14, 2
    streams = [make_streams(S, n, seed0=11), make_streams(S, n, seed0=31)]
    ref = []
    for g in range(2):
        sysm = slam.System(slam.make_config(W, H, S, local_mapping=lm))
        p, _ = run(sysm, streams[g], n)
        ref.append((p, [sysm.stats(s) for s in range(S)]))
    # (deferred: both handles submit to ONE local-BA service, which solves whatever has been submitted as one batch — the results may not depend on that)
    systems = [slam.System(slam.make_config(W, H, S, host_threads=2, local_mapping=lm)) for _ in range(2)]
    out = [None, None]

    def work(g):
        out[g] = run(systems[g], streams[g], n)[0]

    ths = [threading.Thread(target=work, args=(g,)) for g in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for g in range(2):
        assert np.array_equal(out[g], ref[g][0])
        assert [systems[g].stats(s) for s in range(S)] == ref[g][1]


def test_hip_driver_with_lens_distortion_matches_oracle_driver(oracle):
    """Camera.k1..k3 of the reference's Examples/RGB-D/TUM2.yaml: Frame::UndistortKeyPoints and the undistorted image bounds are in the loop."""
    n = 12
    dist = [0.231222, -0.784899, -0.003257, -0.000105, 0.917205]
    streams = make_streams(1, n)
    cfg = slam.make_config(W, H, 1, dist=dist)
    hip = slam.System(cfg)
    ph, sh = run(hip, streams, n)
    cfg_o = slam.make_config(W, H, 1, dist=dist)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run(ora, streams, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all()
    assert hip.stats(0) == ora.stats(0)
    assert np.abs(ph - po).max() < 2e-4


@pytest.mark.parametrize("lm", [slam.LM_SYNC, slam.LM_DEFERRED], ids=["sync", "deferred"])
def test_hip_driver_keyframe_culling_matches_oracle_driver(oracle, lm):
    """Occluded stream (synth.make_occluded_stream): keyframes are inserted in mapped territory and LocalMapping::KeyFrameCulling removes
    redundant ones; the HIP operator table must lead the driver to the same culling decisions as the CPU oracle's table."""
    from object_slam_amd import synth
    n, S = 54, 2
    streams = [synth.make_occluded_stream(n, W, H, seed=sd) for sd in (11, 14)]
    cfg = slam.make_config(W, H, S, local_mapping=lm)
    hip = slam.System(cfg)
    ph, sh = run(hip, streams, n)
    cfg_o = slam.make_config(W, H, S, local_mapping=lm)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run(ora, streams, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all()
    for s in range(S):
        a, b = hip.stats(s), ora.stats(s)
        assert a == b, (s, a, b)
        assert a["map_violations"] == 0, a
        if lm == slam.LM_SYNC:
            assert a["keyframes_culled"] >= 2, a
        assert np.array_equal(hip.keyframe_trajectory(s)[0], ora.keyframe_trajectory(s)[0])
    # (deferred: the culling of pass t sees the map after frame t+1 was tracked; on these streams it then removes fewer keyframes — both tables the same ones)
    assert sum(hip.stats(s)["keyframes_culled"] for s in range(S)) >= 2
    d = np.abs(ph - po).max()
    assert d < 2e-4, d


@pytest.mark.parametrize("lm", [slam.LM_SYNC, slam.LM_DEFERRED], ids=["sync", "deferred"])
def test_hip_driver_matches_oracle_driver_on_3d_scene(oracle, lm):
    """S1 as SURVEY.md §8(d) specifies it: SE3 path with rotation and forward / backward motion over walls and boxes at 1.3 - 4.6 m, so matches
    change pyramid level, triangulation fires and the local BA windows are not planar."""
    from slam_common import ate_scene, make_scene_streams, run_scene
    n, S = 36, 2
    seqs = make_scene_streams(S, n)
    cfg = slam.make_config(W, H, S, local_mapping=lm)
    hip = slam.System(cfg)
    ph, sh = run_scene(hip, seqs, n)
    cfg_o = slam.make_config(W, H, S, local_mapping=lm)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run_scene(ora, seqs, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all()
    for s in range(S):
        a, b = hip.stats(s), ora.stats(s)
        assert a == b, (s, a, b)
        assert a["map_violations"] == 0 and a["local_bas"] >= 2 and a["points_triangulated"] > 0, a
        ah, Th = ate_scene(hip, seqs, s)
        ao, To = ate_scene(ora, seqs, s)
        assert ah < 0.03 and ao < 0.03, (ah, ao)
        assert np.abs(Th - To).max() < 4e-4, np.abs(Th - To).max()      # 1e-4 relative on a scene of up to 4.6 m
    assert np.abs(ph - po).max() < 4e-4, np.abs(ph - po).max()


def test_hip_stereo_driver_matches_oracle_driver_on_street_scene(oracle):
    """S3 shape with forward motion (0.35 m per frame) along a street of facades at 5 - 60 m: stereo depth varies per keypoint."""
    from slam_common import make_scene_stereo, run_scene_stereo, stereo_config
    n, S = 14, 2
    seqs = make_scene_stereo(S, n)
    cfg = stereo_config(S)
    hip = slam.System(cfg)
    ph, sh = run_scene_stereo(hip, seqs, n)
    cfg_o = stereo_config(S)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run_scene_stereo(ora, seqs, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all()
    for s in range(S):
        a, b = hip.stats(s), ora.stats(s)
        assert a == b, (s, a, b)
        assert a["map_violations"] == 0 and a["keyframes_created"] >= 2
    assert np.abs(ph - po).max() < 6e-3, np.abs(ph - po).max()      # 1e-4 relative on a scene of up to 60 m


def _semantic_run(system, q, n, on_device=False):
    import torch
    poses = []
    for t in range(n):
        if on_device:
            g = torch.from_numpy(q["gray"][t]).cuda(); d = torch.from_numpy(q["depth"][t]).cuda(); m = torch.from_numpy(q["masks"][t]).cuda()
            torch.cuda.synchronize()
            objs = [dict(masks=[m[o].data_ptr() for o in range(3)], track_ids=q["track_ids"])]
            T, st = system.TrackRGBD([g.data_ptr()], [d.data_ptr()], [t / 30.0], objects=objs, on_device=True, gray_stride=W, depth_pitch=W, mask_stride=W)
        else:
            objs = [dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])]
            T, st = system.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
        assert st[0] == slam.OK, t
        poses.append(T[0].copy())
    return np.array(poses)


def test_hip_semantic_tracking_matches_oracle_driver(oracle):
    """BASELINE.json configs[2]: instance masks in, Frame::BuildObject2DsRGBD + ObjectOptimizer::PoseOptimization2 inside TrackLocalMap (reference
    src/Tracking.cc:1022).  HIP operator table (k_object_kp_test, k_pose_optimize<true> batch form) against the CPU oracle's table: identical object
    bookkeeping and nSemNum totals, poses within the optimiser tolerance; masks on the host and resident in HBM give the same result; the semantic
    edges change the trajectory."""
    from slam_common import make_scene_streams
    n = 24
    q = make_scene_streams(1, n)[0]
    hip = slam.System(slam.make_config(W, H, 1))
    ph = _semantic_run(hip, q, n)
    cfg_o = slam.make_config(W, H, 1)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po = _semantic_run(ora, q, n)
    a, b = hip.stats(0), ora.stats(0)
    assert a == b, (a, b)
    assert a["semantic_frames"] == n - 2 and a["semantic_frames_nonzero"] == n - 2 and a["object3ds"] == 3 and a["semantic_edges"] > 100 * (n - 2), a
    assert np.abs(ph - po).max() < 4e-4, np.abs(ph - po).max()
    dev = slam.System(slam.make_config(W, H, 1))
    pd = _semantic_run(dev, q, n, on_device=True)
    assert np.array_equal(pd, ph) and dev.stats(0) == a
    plain = slam.System(slam.make_config(W, H, 1))
    pp = np.array([plain.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0])[0][0].copy() for t in range(n)])
    assert np.abs(pp - ph).max() > 1e-5


def test_raw16_depth_and_bit_masks_in_pinned_host_memory_equal_the_float_run():
    """oslam_slam_track_rgbd_raw16: raw 16-bit depth images scaled on lookup (DepthMapFactor, reference src/Tracking.cc:262) and caller-packed one-bit masks,
    all in PINNED HOST memory read by the kernels over PCIe, give bit-identical poses and statistics to the run on float depth images and byte masks;
    raw16 depth in pageable host memory (scaled while staging) too."""
    import torch
    from object_slam_amd import seqbench
    from object_slam_amd.scene import DEPTH_FACTOR, depth_to_metres
    from slam_common import make_scene_streams
    n = 16
    q = make_scene_streams(1, n)[0]
    ref = slam.System(slam.make_config(W, H, 1))
    pr = _semantic_run(ref, q, n)
    d16 = np.rint(q["depth"].astype(np.float64) * DEPTH_FACTOR).astype(np.uint16)
    assert np.array_equal(depth_to_metres(d16), q["depth"])
    bits = seqbench.pack_mask_bits(q["masks"])                       # [n, 3, H, 10] uint64
    assert bits.shape == (n, 3, H, (W + 63) // 64)
    g_p, d_p, b_p = torch.from_numpy(q["gray"]).pin_memory(), torch.from_numpy(d16).pin_memory(), torch.from_numpy(bits.view(np.int64)).pin_memory()
    factor = float(np.float32(1.0) / np.float32(DEPTH_FACTOR))
    raw = slam.System(slam.make_config(W, H, 1))

    def tab(t_, per):
        return (t_.data_ptr() + np.arange(n, dtype=np.uint64) * np.uint64(per)).reshape(n, 1)
    masks = np.stack([tab(b_p, 3 * H * 10 * 8) + np.uint64(o * H * 10 * 8) for o in range(3)], 2)
    calls = raw.prepare_rgbd_bulk(tab(g_p, W * H), tab(d_p, W * H * 2), np.arange(n, dtype=np.float64).reshape(n, 1) / 30.0, W, W, masks=masks,
                                  track_ids=np.array([q["track_ids"]], np.int32), labels=np.zeros((1, 3), np.int32), depth_u16_factor=factor, mask_bits=True, on_device=1)
    pw = np.array([raw.track_prepared(c)[0][0].copy() for c in calls])
    assert np.array_equal(pw, pr) and raw.stats(0) == ref.stats(0)
    # pageable host memory, byte masks, raw 16-bit depth
    pag = slam.System(slam.make_config(W, H, 1))
    masks_h = np.stack([(q["masks"].ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(3 * H * W) + np.uint64(o * H * W)).reshape(n, 1) for o in range(3)], 2)
    calls = pag.prepare_rgbd_bulk((q["gray"].ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(W * H)).reshape(n, 1),
                                  (d16.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(W * H * 2)).reshape(n, 1),
                                  np.arange(n, dtype=np.float64).reshape(n, 1) / 30.0, W, W, masks=masks_h, track_ids=np.array([q["track_ids"]], np.int32),
                                  labels=np.zeros((1, 3), np.int32), mask_stride=W, depth_u16_factor=factor, on_device=0)
    pp = np.array([pag.track_prepared(c)[0][0].copy() for c in calls])
    assert np.array_equal(pp, pr) and pag.stats(0) == ref.stats(0)


def test_local_search_capacity_grows_on_demand(oracle, monkeypatch):
    """The reference's local map is unbounded.  With the first reservation of the local-point matcher forced below the size of the local map, the HIP
    table re-creates the matcher instead of failing the step; the run equals the oracle table's."""
    from slam_common import make_scene_streams, run_scene
    monkeypatch.setenv("OSLAM_SLAM_MAX_LOCAL", "256")
    n = 10
    seqs = make_scene_streams(1, n)
    hip = slam.System(slam.make_config(W, H, 1))
    ph, sh = run_scene(hip, seqs, n)
    cfg_o = slam.make_config(W, H, 1)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run_scene(ora, seqs, n)
    assert np.array_equal(sh, so) and (sh == slam.OK).all() and hip.stats(0) == ora.stats(0)
    assert np.abs(ph - po).max() < 4e-4


def test_lost_and_reset_on_one_slot_matches_oracle_driver(oracle):
    """Sequence 0 starts on a textureless frame (NOT_INITIALIZED), tracks, loses the track on two blank frames with <= 5 keyframes (System::Reset,
    reference src/Tracking.cc:553-561) and builds a new map whose keyframe ids restart at 0; sequence 1 runs undisturbed in the neighbouring slot.
    The HIP table's per-slot resident state (keyframe records, local-map arrays) must follow the reset: same states, statistics and poses as the
    oracle table."""
    from slam_common import make_scene_streams
    n = 16
    seqs = make_scene_streams(2, n, speed=1.5)
    blank = np.full((H, W), 90, np.uint8)
    flat = np.full((H, W), 2.0, np.float32)
    g0 = [blank] + list(seqs[0]["gray"][:6]) + [blank, blank] + list(seqs[0]["gray"][6:13])
    d0 = [flat] + list(seqs[0]["depth"][:6]) + [flat, flat] + list(seqs[0]["depth"][6:13])

    def run(system):
        poses, states = [], []
        for t in range(n):
            T, st = system.TrackRGBD([g0[t], seqs[1]["gray"][t]], [d0[t], seqs[1]["depth"][t]], [t / 30.0] * 2)
            poses.append(T.copy()); states.append(st.copy())
        return np.array(poses), np.array(states)

    hip = slam.System(slam.make_config(W, H, 2))
    ph, sh = run(hip)
    cfg_o = slam.make_config(W, H, 2)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po, so = run(ora)
    assert np.array_equal(sh, so), (sh.T, so.T)
    assert sh[0, 0] == slam.NOT_INITIALIZED and (sh[1:7, 0] == slam.OK).all() and sh[7, 0] == slam.LOST and sh[8, 0] == slam.NOT_INITIALIZED
    assert (sh[9:, 0] == slam.OK).all() and (sh[:, 1] == slam.OK).all()
    assert hip.stats(0) == ora.stats(0) and hip.stats(1) == ora.stats(1)
    assert hip.stats(0)["lost_frames"] == 1 and hip.stats(0)["map_violations"] == 0
    assert np.abs(ph - po).max() < 4e-4
    assert len(hip.trajectory(0)[0]) == n - 9           # the trajectory of the new map only


def test_resident_map_point_records_equal_the_host_map():
    """The HIP table keeps a 64-byte record per map point (position, normal, distances, descriptor) that every MapPoint update writes.  After a run with
    keyframe insertions, triangulation, fusions, local BA and cullings the record of EVERY point that is alive must equal the driver's host copy bit for bit."""
    from slam_common import make_scene_streams, run_scene
    n = 40
    seqs = make_scene_streams(2, n, speed=2.0)
    hip = slam.System(slam.make_config(W, H, 2))
    run_scene(hip, seqs, n)
    for s in range(2):
        st = hip.stats(s)
        assert st["local_bas"] >= 3 and st["points_fused"] > 0 and st["points_triangulated"] > 0
        checked = 0
        for pid in range(st["points_created"]):
            host, res, bad = hip.debug_point(s, pid)
            assert np.array_equal(host[:12], res[:12]), (s, pid, bad)   # the position of EVERY point, culled ones included (object lists keep them)
            if bad:
                continue
            assert np.array_equal(host, res), (s, pid, host.view(np.float32)[:8], res.view(np.float32)[:8])
            checked += 1
        assert checked >= st["points_in_map"] > 500


def test_resident_local_maps_and_keyframes_leave_the_run_unchanged(monkeypatch):
    """The HIP table keeps the packed SearchLocalPoints arrays of every slot, the keyframes' arrays and a record per map point resident in HBM (content ids /
    keyed operators / jobs by map-point id); with all of it switched off it uploads the host arrays of every job as the oracle table receives them.  The two runs must be bit-identical, and the
    driver must actually have reused local maps in the first one (slow motion: most frames keep their local keyframe list)."""
    from slam_common import make_scene_streams, run_scene
    n = 24
    seqs = make_scene_streams(2, n, speed=1.0)
    a = slam.System(slam.make_config(W, H, 2))
    pa, sa = run_scene(a, seqs, n)
    reused, frames = a.local_map_reuse()
    assert frames >= 2 * (n - 2) and reused >= frames // 3, (reused, frames)
    monkeypatch.setenv("OSLAM_SLAM_NO_RESIDENT_LOCAL", "1")
    monkeypatch.setenv("OSLAM_SLAM_NO_RESIDENT_KF", "1")
    monkeypatch.setenv("OSLAM_SLAM_NO_RESIDENT_POINTS", "1")   # pose / search_last / Fuse jobs carry their arrays again instead of map-point ids
    b = slam.System(slam.make_config(W, H, 2))
    pb, sb = run_scene(b, seqs, n)
    assert np.array_equal(sa, sb) and np.array_equal(pa, pb)
    assert a.stats(0) == b.stats(0) and a.stats(1) == b.stats(1)


def test_create_new_map_points_in_one_batch_equals_the_neighbour_rounds(monkeypatch):
    """LocalMapping::CreateNewMapPoints searches and triangulates ALL neighbours of a keyframe in one batch from the state before the pass and applies the results
    in neighbour order with the reference's skip test (a keypoint that has received a point) at application time; OSLAM_SLAM_CNMP_ROUNDS=1 keeps one lockstep round
    per neighbour.  Both give the same points in the same order: bit-identical poses and statistics over a run that triangulates."""
    from slam_common import make_scene_streams, run_scene
    n, S = 40, 2
    seqs = make_scene_streams(S, n)
    a = slam.System(slam.make_config(W, H, S))
    pa, sa = run_scene(a, seqs, n)
    monkeypatch.setenv("OSLAM_SLAM_CNMP_ROUNDS", "1")
    b = slam.System(slam.make_config(W, H, S))
    pb, sb = run_scene(b, seqs, n)
    assert np.array_equal(sa, sb) and np.array_equal(pa, pb)
    for q in range(S):
        st = a.stats(q)
        assert st == b.stats(q) and st["points_triangulated"] > 50, st


def test_hip_driver_200_frames_with_masks_matches_oracle_driver(oracle):
    """Soak at the S1 specification: 200 frames at speed 1 (<= 2 cm, <= 0.5 deg per frame) with the three instance masks; the HIP and
    the oracle operator tables must lead the driver through the same 200 frames (states, map statistics, object bookkeeping), ATE below 2 cm."""
    from slam_common import ate_scene
    from object_slam_amd import scene
    n = 200
    q = scene.make_rgbd_sequence(5, n, speed=1.0)
    hip = slam.System(slam.make_config(W, H, 1))
    ph = _semantic_run(hip, q, n)
    cfg_o = slam.make_config(W, H, 1)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    po = _semantic_run(ora, q, n)
    a, b = hip.stats(0), ora.stats(0)
    assert a == b, (a, b)
    assert a["lost_frames"] == 0 and a["map_violations"] == 0 and a["keyframes_created"] >= 8 and a["local_bas"] >= 6 and a["semantic_frames_nonzero"] >= n - 3, a
    assert np.abs(ph - po).max() < 4e-4, np.abs(ph - po).max()
    ah, _ = ate_scene(hip, [q], 0)
    assert ah < 0.02, ah
    # the run reaches observations in culled keyframes (left out of ComputeDistinctiveDescriptors, reference src/MapPoint.cc:366): both tables saw the same
    assert hip.bad_keyframe_observations() == ora.bad_keyframe_observations() > 0


def test_hip_driver_200_frames_with_masks_deferred_schedule_tracks_the_oracle_driver(oracle):
    """The same soak under the deferred local-mapping schedule, both tables advanced frame by frame.  The tables agree on every statistic and within the optimiser
    tolerance on every pose for as long as no thresholded decision falls inside that tolerance: on this stream the first such decision is a triangulation test at
    frame 127 (one more point passes on one side: poses 6e-6 apart at that frame, 4e-5 at most before it), after which the two maps legitimately differ.  Required:
    identical statistics and poses within 4e-4 on at least the first 100 frames, the first difference a single unit of a counter, and both runs healthy to the
    end (no lost frame, no map violation, ATE below 2 cm, the same number of keyframes within 10 %)."""
    from slam_common import ate_scene
    from object_slam_amd import scene
    n = 200
    q = scene.make_rgbd_sequence(5, n, speed=1.0)
    hip = slam.System(slam.make_config(W, H, 1, local_mapping=slam.LM_DEFERRED))
    cfg_o = slam.make_config(W, H, 1, local_mapping=slam.LM_DEFERRED)
    ora = slam.System(cfg_o, oracle_ops(cfg_o))
    first_diff = None
    for t in range(n):
        objs = [dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])]
        Th, sh = hip.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
        To, so = ora.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
        assert sh[0] == slam.OK and so[0] == slam.OK, t
        if first_diff is None:
            a, b = hip.stats(0), ora.stats(0)
            diff = {k: (a[k], b[k]) for k in a if a[k] != b[k]}
            if diff:
                first_diff = t
                assert all(abs(x - y) <= 2 for x, y in diff.values()), (t, diff)      # one decision (a point counts once in created / in map / triangulated, twice in lba_edges)
            else:
                assert np.abs(Th[0] - To[0]).max() < 4e-4, (t, np.abs(Th[0] - To[0]).max())
    assert first_diff is None or first_diff >= 100, first_diff
    a, b = hip.stats(0), ora.stats(0)
    for st in (a, b):
        assert st["lost_frames"] == 0 and st["map_violations"] == 0 and st["keyframes_created"] >= 8 and st["local_bas"] >= 6, st
    assert abs(a["keyframes_created"] - b["keyframes_created"]) <= 0.1 * b["keyframes_created"], (a, b)
    ah, _ = ate_scene(hip, [q], 0)
    ao, _ = ate_scene(ora, [q], 0)
    assert ah < 0.02 and ao < 0.02, (ah, ao)


# ---- the C++ driver over the HIP operators against the INDEPENDENT Python restatement of the driver ----
# tests/golden/driver_*.npz: per-frame states, poses and map statistics of oracle/slam_driver_oracle.py over the CPU oracle's operators, generated in the build
# container by tests/golden/gen_driver_golden.py (the Python driver takes minutes; the fixtures travel as data).  Every discrete statistic must agree on every
# frame; poses within the optimisers' 1e-4 relative tolerance (2e-4 absolute on these ~2 m scenes).
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


@pytest.mark.parametrize("tag,lm", [("sync", slam.LM_SYNC), ("deferred", slam.LM_DEFERRED)])
def test_hip_driver_matches_python_restatement_fixture(tag, lm):
    g = _golden("driver_rgbd_%s.npz" % tag)
    n = int(g["n"])
    assert int(g["local_mapping"]) == lm
    streams = make_streams(1, n)
    depth = np.full((H, W), 2.0, np.float32)
    sysm = slam.System(slam.make_config(W, H, 1, local_mapping=lm))
    keys = [str(k) for k in g["stat_keys"]]
    for t in range(n):
        T, st = sysm.TrackRGBD([streams[0][0][t]], [depth], [t / 30.0])
        assert int(st[0]) == int(g["states"][t]), t
        assert np.abs(T[0] - g["poses"][t]).max() < 2e-4, (t, np.abs(T[0] - g["poses"][t]).max())
        s = sysm.stats(0)
        got = [int(s[k]) for k in keys]
        assert got == [int(v) for v in g["stats"][t]], (t, dict(zip(keys, got)), dict(zip(keys, g["stats"][t])))
    sysm.finish()
    _, Twc = sysm.trajectory(0)
    assert Twc.shape == g["trajectory"].shape and np.abs(Twc - g["trajectory"]).max() < 2e-4
    assert sysm.stats(0)["map_violations"] == 0 and sysm.stats(0)["local_bas"] >= 3


def test_hip_semantic_driver_matches_python_restatement_fixture():
    """BASELINE.json configs[2] shape (three box objects with instance masks, ObjectOptimizer::PoseOptimization2 in TrackLocalMap) against the Python restatement."""
    from slam_common import make_scene_streams
    g = _golden("driver_semantic_sync.npz")
    n = int(g["n"])
    q = make_scene_streams(1, n)[0]
    sysm = slam.System(slam.make_config(W, H, 1))
    keys = [str(k) for k in g["stat_keys"]]
    for t in range(n):
        objs = dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])
        T, st = sysm.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=[objs])
        assert int(st[0]) == int(g["states"][t]), t
        assert np.abs(T[0] - g["poses"][t]).max() < 2e-4, (t, np.abs(T[0] - g["poses"][t]).max())
        s = sysm.stats(0)
        got = [int(s[k]) for k in keys]
        assert got == [int(v) for v in g["stats"][t]], (t, dict(zip(keys, got)), dict(zip(keys, g["stats"][t])))
    assert sysm.stats(0)["semantic_edges"] > 1000 and sysm.stats(0)["object3ds"] == 3


@pytest.mark.parametrize("lm", [slam.LM_SYNC, slam.LM_DEFERRED], ids=["sync", "deferred"])
def test_operator_failure_is_confined_to_its_sequence(lm):
    """Per-sequence failure isolation (include/oslam_slam.h; reference: one System resets itself, src/Tracking.cc:553-560): the local-BA window of ONE sequence of a
    4-sequence handle is made invalid (oslam_slam_inject_failure) — that sequence reports LOST, starts a new map and is counted; the other three sequences come
    out bit-identical to a run without the failure."""
    n, S, victim = 30, 4, 2
    streams = make_streams(S, n)
    clean = slam.System(slam.make_config(W, H, S, local_mapping=lm))
    pc, sc = run(clean, streams, n)
    hit = slam.System(slam.make_config(W, H, S, local_mapping=lm))
    depth = np.full((H, W), 2.0, np.float32)
    poses, states = [], []
    injected_at = None
    for t in range(n):
        if injected_at is None and hit.stats(victim)["local_bas"] >= 1:     # after its first local BA: the next window of the victim is refused
            hit.inject_failure(victim)
            injected_at = t
        T, st = hit.TrackRGBD([streams[s][0][t] for s in range(S)], [depth] * S, [t / 30.0] * S)
        poses.append(T.copy()); states.append(st.copy())
    hit.finish(); clean.finish()
    poses, states = np.array(poses), np.array(states)
    assert injected_at is not None
    others = [s for s in range(S) if s != victim]
    assert np.array_equal(poses[:, others], pc[:, others]) and np.array_equal(states[:, others], sc[:, others])
    for s in others:
        assert hit.stats(s) == clean.stats(s) and hit.lba_window_stats(s)["operator_failures"] == 0
    w = hit.lba_window_stats(victim)
    assert w["operator_failures"] == 1
    # the failure is visible to the caller: the sequence re-initialises on the frame after the refused window (StereoInitialization: the identity pose of a new map) ...
    eye = np.eye(4, dtype=np.float32)
    restarts = [t for t in range(injected_at, n) if np.array_equal(poses[t, victim], eye)]
    assert len(restarts) == 1, restarts
    # ... and tracks again on that map
    assert states[-1, victim] == slam.OK and hit.stats(victim)["map_violations"] == 0 and hit.stats(victim)["lost_frames"] == 0
    assert np.array_equal(poses[:injected_at, victim], pc[:injected_at, victim])


def test_device_culling_counts_equal_the_host_walk():
    """KeyFrameCulling's redundancy counts come from the device's mirror of the observation graph (include/oslam_slam.h: map_journal / kf_culling_counts).
    OSLAM_SLAM_CULL_CHECK=1 makes the driver recount every candidate on the host and abort on a difference; OSLAM_SLAM_CULL_HOST=1 is the host walk alone.
    Both runs (occluded streams: keyframes ARE culled) must end with the same statistics and bit-identical poses."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    # (third run: WITHOUT the deferred form of the per-round descriptor updates of SearchInNeighbors, oslam_slam_ops_t::mp_update_keyed_async — the default since round 5)
    # (the same two processes cover SearchInNeighbors' second direction from the mirror — oslam_slam_ops_t::fuse_into_current: OSLAM_SLAM_FUSECUR_CHECK=1 compares the
    # table's candidate list and flags with the driver's entry by entry, OSLAM_SLAM_FUSECUR_HOST=1 is the driver's own walk — and, in the check process, the opt-in
    # Tracking::UpdateLocalPoints from the mirror: oslam_slam_ops_t::local_points_list, OSLAM_SLAM_LOCLIST_DEV=1, compared list by list)
    for tag, env in (("check", {"OSLAM_SLAM_CULL_CHECK": "1", "OSLAM_SLAM_FUSECUR_CHECK": "1", "OSLAM_SLAM_LOCLIST_DEV": "1", "OSLAM_SLAM_LOCLIST_CHECK": "1"}), ("host", {"OSLAM_SLAM_CULL_HOST": "1", "OSLAM_SLAM_FUSECUR_HOST": "1"}),
                     ("mpu_sync", {"OSLAM_SLAM_MPU_SYNC": "1"})):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.join(here, "cull_check_run.py")], env=e, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (tag, p.stdout[-2000:], p.stderr[-2000:])
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        res[tag] = json.loads(line[7:])
    assert res["check"] == res["host"] == res["mpu_sync"]
    for name in ("sync", "deferred"):
        assert res["check"][name]["status_ok"]
    assert sum(s["keyframes_culled"] for s in res["check"]["sync"]["stats"]) >= 2
