"""Multi-GPU sharding of independent streams.

Frames shard by stream: rank r filters stream r on GPU r; there is no data-path
collective.  The only communication is the final throughput reduction
(frames: SUM, seconds: MAX) over torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in the CPU tests)."""
from __future__ import annotations


def stream_for_rank(rank: int, world: int, n_streams: int):
    """Indices of the streams rank `rank` owns (round-robin, whole streams only:
    hqdn3d / EEDI2 carry state across frames, so a stream is never split)."""
    return list(range(rank, n_streams, world))


def reduce_throughput(frames_local: float, seconds_local: float, device=None):
    """(total frames, max seconds) over all ranks; identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(frames_local), float(seconds_local)
    kw = {} if device is None else {"device": device}
    f = torch.tensor([frames_local], dtype=torch.float64, **kw)
    t = torch.tensor([seconds_local], dtype=torch.float64, **kw)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item()), float(t.item())
