"""Multi-GPU layout of the generation path: utterances are independent, so ranks take contiguous shards of the batch
and NO data-path collective exists.  torch.distributed is only used to agree on the wall time of a run
(max over ranks) and, optionally, to gather the int/float sample arrays on rank 0."""


def shard_range(n_items, world_size, rank):
    """[start, stop) of rank's contiguous shard; the first n_items % world_size ranks take one extra item."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (identity without an initialised process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_on_rank0(array, device=None):
    """rank 0 receives the list of every rank's numpy array (others get None); shards may be ragged."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [array]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(array, out, dst=0)
    return out
