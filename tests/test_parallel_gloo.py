"""CPU, world_size 2 over gloo: the batch-of-sequences entry point bench.py runs on every rank (`seqbench.run_rank`: sharding,
barriers around the timed region, max-over-ranks time, all-gather of the per-rank record) — here over the CPU oracle's operator table
instead of the HIP one (RCCL on the GPU box)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from object_slam_amd.parallel import aggregate_stats, gather_records, shard_sequences


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_system(cfg):
    import ctypes as C

    from object_slam_amd import slam
    from oracle import oracle_py as O
    ops = slam.SlamOps()
    assert O.lib().oo_slam_make_ops(C.byref(cfg), C.byref(ops)) == 0
    return slam.System(cfg, ops)


def _worker(rank, world, port, q, kind="rgbd"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from object_slam_amd import seqbench
    if kind == "stereo":   # BASELINE.json configs[4]: KITTI-shaped stereo sequences dealt to the ranks
        wl = seqbench.stereo_workload(n_base=2, stagger=1)
        summ, rec, systems, extra = seqbench.run_rank(wl, _oracle_system, rank, world, 2, 2, steps=2, warmup=1, on_device=False, collect_poses=True)
    else:
        wl = seqbench.rgbd_workload(n_base=2, stagger=1)
        from object_slam_amd import slam
        summ, rec, systems, extra = seqbench.run_rank(wl, _oracle_system, rank, world, 2, 2, steps=3, warmup=2, on_device=False, collect_poses=True,
                                                      local_mapping=slam.LM_DEFERRED)   # bench.py's default schedule (the stereo test below runs the synchronous one)
    poses = np.array(extra["poses"])          # [handles, frames, 1, 4, 4]
    q.put((rank, summ, rec.tolist(), poses, extra["groups"]))
    if world > 1:
        dist.destroy_process_group()


def _run(world, kind="rgbd"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_run_rank_world2_over_gloo(oracle):
    res = _run(2)
    s0, s1 = res[0][1], res[1][1]
    # every rank sees the same whole-job summary: frames summed, time = max over ranks
    assert s0["total_frames"] == s1["total_frames"] == 2 * 2 * 3 and s0["n_ranks"] == 2
    assert s0["elapsed_s"] == s1["elapsed_s"] and s0["frames_per_s"] == s1["frames_per_s"]
    rec = np.array(res[0][2])
    assert np.array_equal(rec, np.array(res[1][2])) and rec.shape[0] == 2
    assert list(rec[:, 0]) == [0, 1] and list(rec[:, 2]) == [6, 6]
    assert s0["elapsed_s"] == max(rec[:, 3])
    assert res[0][4] == [[0], [2]] and res[1][4] == [[1], [3]]        # sequence i -> rank i mod 2, dealt to the rank's handles in order
    assert s0["lost_frames"] == 0 and s0["map_violations"] == 0 and s0["keyframes"] >= 4
    # the shard a rank runs does not depend on the world size: rank 0 of the 2-rank job = the same sequences run alone
    one = _run(1)
    assert one[0][4] == [[0], [1]]
    # sequence 0 is in both runs (same base sequence and offset): identical poses
    assert np.array_equal(one[0][3][0], res[0][3][0])


def test_run_rank_world2_over_gloo_stereo(oracle):
    """configs[4] shape: the stereo workload (1241x376, 2000 features) through the same entry point at world_size 2."""
    res = _run(2, "stereo")
    s0, s1 = res[0][1], res[1][1]
    assert s0["total_frames"] == s1["total_frames"] == 2 * 2 * 2 and s0["n_ranks"] == 2
    assert s0["elapsed_s"] == s1["elapsed_s"] and s0["frames_per_s"] == s1["frames_per_s"]
    rec = np.array(res[0][2])
    assert np.array_equal(rec, np.array(res[1][2])) and rec.shape[0] == 2
    assert list(rec[:, 0]) == [0, 1] and list(rec[:, 2]) == [4, 4]
    assert res[0][4] == [[0], [2]] and res[1][4] == [[1], [3]]
    assert s0["lost_frames"] == 0 and s0["map_violations"] == 0 and s0["keyframes"] >= 4      # every sequence initialises from its first stereo pair
    # rank 0's sequence 0 alone (world 1) gives the same poses: the shard does not depend on the world size
    one = _run(1, "stereo")
    assert np.array_equal(one[0][3][0], res[0][3][0])


def test_shard_and_aggregate_passthrough():
    assert shard_sequences(5, 1, 0) == [0, 1, 2, 3, 4]
    assert shard_sequences(8, 2, 1) == [1, 3, 5, 7]
    assert aggregate_stats(2.0, 10) == (10, 2.0)
    assert gather_records([1.0, 2.0]).tolist() == [[1.0, 2.0]]


def _worker_unequal(rank, world, port, q, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from object_slam_amd import seqbench
    wl = seqbench.rgbd_workload(n_base=2, stagger=1, with_masks=False)
    summ, rec, systems, extra = seqbench.run_rank(wl, _oracle_system, rank, world, 0, 2, steps=2, warmup=1, on_device=False, collect_poses=True, total_sequences=total)
    q.put((rank, summ, rec.tolist(), extra["groups"], [np.array(p) for p in extra["poses"]]))
    dist.destroy_process_group()


def test_run_rank_world4_with_unequal_shards(oracle):
    """6 sequences on 4 ranks (the sequence count is not a multiple of the world size: configs[4] names 8 sequences on 1 / 2 / 4 / 8 GPUs, a deployment any count):
    sequence i -> rank i mod 4, so ranks 0 and 1 run two sequences (one per handle) and ranks 2 and 3 one; frames are summed over the ranks, the time is the
    slowest rank's, and a sequence's poses do not depend on the rank or the handle it ran in."""
    world, total = 4, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_unequal, args=(r, world, port, q, total)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [[[0], [4]], [[1], [5]], [[2]], [[3]]]
    rec = np.array(res[0][2])
    assert all(np.array_equal(rec, np.array(r[2])) for r in res) and rec.shape[0] == 4
    assert list(rec[:, 1]) == [2, 2, 1, 1] and list(rec[:, 2]) == [4, 4, 2, 2]          # sequences and timed frames per rank
    s = res[0][1]
    assert s["total_frames"] == total * 2 and s["n_ranks"] == 4 and s["elapsed_s"] == max(rec[:, 3])
    assert all(r[1]["frames_per_s"] == s["frames_per_s"] for r in res) and s["lost_frames"] == 0 and s["map_violations"] == 0
    # every handle of every rank advanced its sequences through warm-up + timed steps from the identity pose of a new map
    for r in res:
        for p in r[4]:
            assert p.shape[0] == 3 and p.shape[-2:] == (4, 4) and np.array_equal(p[0, 0], np.eye(4, dtype=np.float32))


def _shared_worker(local_rank, local_world, tag, shm_dir, q):
    from object_slam_amd import seqbench
    wl = seqbench.rgbd_workload(n_base=2, stagger=1, with_masks=True)
    seqs = seqbench.shared_base_sequences(wl, local_rank, local_world, 2, 3, tag, workers=1, shm_dir=shm_dir, timeout_s=120.0)
    import hashlib
    dig = {b: {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in s.items() if isinstance(v, np.ndarray)} for b, s in seqs.items()}
    mapped = {b: isinstance(s["gray"], np.memmap) for b, s in seqs.items()}
    q.put((local_rank, dig, mapped, {b: int(s["stream_seed"]) for b, s in seqs.items()}))


def test_base_streams_shared_between_the_ranks_of_a_node(tmp_path):
    """One segment for the node's rendered base streams (seqbench.shared_base_sequences, bench.py OSLAM_BENCH_SHARED_BASES=1): local rank 0 renders, the others
    map; every rank sees rank 0's streams bit for bit, the images are file mappings (one copy for the node) and no file is left behind."""
    from object_slam_amd import seqbench
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_shared_worker, args=(r, world, "t%d" % os.getpid(), str(tmp_path), q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    wl = seqbench.rgbd_workload(n_base=2, stagger=1, with_masks=True)
    import hashlib
    ref = seqbench.base_sequences(wl, 0, 2, 3)
    ref_dig = {b: {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in s.items() if isinstance(v, np.ndarray)} for b, s in ref.items()}
    for r in res:
        assert r[1] == ref_dig and all(r[2].values()) and r[3] == {0: 0, 1: 1}
    assert os.listdir(str(tmp_path)) == []
