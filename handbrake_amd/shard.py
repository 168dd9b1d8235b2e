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


def reduce_host_path(res: dict, device=None) -> dict:
    """Every rank ran the PCIe-inclusive pass (handbrake_amd/hostpath.py) at the same time, each on its own GPU: the
    job's rate is all ranks' credited output frames over the slowest rank's time - what the host's PCIe root complex
    and memory give N streams at once.  `res` = this rank's result (n_out, seconds, value, ...); returns
    it with `value` replaced by the aggregate, the rank's own rate kept as `this_rank_value`."""
    import torch.distributed as dist
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    n_all, dt_all = reduce_throughput(float(res.get("n_out", 0)), float(res.get("seconds", 0.0)), device=device)
    out = dict(res)
    if world > 1 and "error" not in res:
        out["this_rank_value"] = res.get("value")
        out["value"] = round(n_all / dt_all, 2) if dt_all > 0 else None
        out["ranks"] = world
    return out
