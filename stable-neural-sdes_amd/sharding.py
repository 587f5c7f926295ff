"""Row sharding for one-process-per-GPU runs: every batch row is an independent SDE, so a
solve shards by contiguous row ranges with NO collective on the data path.  Only the bookkeeping lives
here: which rows a rank owns (and therefore its Philox ``row_offset``) and the max-over-ranks timing
reduction used by bench.py."""
import torch


def shard_rows(n_rows, world_size, rank):
    """Contiguous, balanced row range [lo, hi) of `rank`; lo is the shard's global Philox row_offset."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, extra = divmod(n_rows, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (no-op without an initialised process group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, n_rows, dim=0):
    """All-gather row shards of unequal size along `dim` (utility for tests / evaluation)."""
    import torch.distributed as dist
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, local.cpu())
    out = torch.cat(parts, dim=dim)
    assert out.shape[dim] == n_rows
    return out
