"""CPU, world_size 2 over gloo: the batch-of-sequences sharding and the end-of-run reduction that
bench.py uses at N>1 (RCCL on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_slam_amd.parallel import aggregate_stats, gather_records, shard_sequences


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_sequences(8, world, rank)
    frames = 100 * len(mine) + rank
    elapsed = 1.0 + 0.5 * rank
    dist.barrier()
    tot, tmax = aggregate_stats(elapsed, frames)
    rec = gather_records([rank, frames, elapsed])
    q.put((rank, mine, tot, tmax, rec.tolist()))
    dist.destroy_process_group()


def test_shard_and_aggregate_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    for r in res:
        assert r[2] == 400 + 401 and abs(r[3] - 1.5) < 1e-12
        assert r[4] == [[0.0, 400.0, 1.0], [1.0, 401.0, 1.5]]


def test_single_process_passthrough():
    assert shard_sequences(5, 1, 0) == [0, 1, 2, 3, 4]
    assert aggregate_stats(2.0, 10) == (10, 2.0)
    assert gather_records([1.0, 2.0]).tolist() == [[1.0, 2.0]]
