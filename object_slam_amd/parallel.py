"""Batch-of-sequences sharding: independent sequences are dealt to ranks, one
process per GPU; there is no exchange inside the hot loop.  torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for the barrier around the timed
region and the final reduction of a fixed-size stats record."""
import torch
import torch.distributed as dist


def shard_sequences(n_sequences, world, rank):
    """Sequence i -> rank i mod world."""
    return [i for i in range(n_sequences) if i % world == rank]


def aggregate_stats(elapsed_s, frames, device=None):
    """Whole-job stats: frames summed over ranks, wall time = max over ranks.
    Returns (total_frames, max_elapsed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(frames), float(elapsed_s)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    f = torch.tensor([int(frames)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return int(f.item()), float(t.item())


def gather_records(record, device=None):
    """All-gather of a fixed-size float64 record per rank (e.g. frames/s, matches, ATE)."""
    r = torch.as_tensor(record, dtype=torch.float64, device=device).reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return r.unsqueeze(0).cpu()
    out = [torch.empty_like(r) for _ in range(dist.get_world_size())]
    dist.all_gather(out, r)
    return torch.stack(out).cpu()
