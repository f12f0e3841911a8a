"""Batch sharding of the path across the GPUs of one node: images are independent units (eval-mode BatchNorm, no
cross-sample op anywhere; decoder.rs:129), so rank r of G owns a contiguous run of the global batch and a full weight
replica; there is no data-path collective.  torch.distributed is used for control only (barrier, max-over-ranks)."""


def shard_range(global_batch: int, world: int, rank: int):
    """[start, stop) of the images rank `rank` owns; the first (global_batch % world) ranks take one extra image."""
    if world < 1 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard arguments")
    q, r = divmod(global_batch, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """max of a per-rank scalar (the timing reduction of bench.py); identity without a process group."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
