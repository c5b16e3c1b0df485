"""Query sharding across the GPUs of one node (one process per GPU) and the single gather of the top-k.

Queries are independent (match_maker.py:192-203 is evaluated per row), so rank r of W owns the contiguous query range
[Q*r/W, Q*(r+1)/W) and the truth index, title tables and word counts are replicated on every GPU.  The only
collective is one all-gather of the int32[Q/W, k] row indexes (RCCL over xGMI when the backend is "nccl"; gloo in the
CPU tests).  torch.distributed is plumbing only; nothing here computes.
"""
import numpy as np


def shard_range(n_queries, rank, world_size):
    """Contiguous query range [begin, end) of `rank`."""
    begin = (n_queries * rank) // world_size
    end = (n_queries * (rank + 1)) // world_size
    return begin, end


def shard_sizes(n_queries, world_size):
    return [shard_range(n_queries, r, world_size)[1] - shard_range(n_queries, r, world_size)[0]
            for r in range(world_size)]


def gather_rows(local_rows, n_queries, group=None):
    """All-gather the per-rank top-k rows (torch tensor int32[q_local, k], on the backend's device) into
    int32[n_queries, k] in query order.  Shards may differ by one row; they are padded to the longest."""
    import torch
    import torch.distributed as dist
    world_size = dist.get_world_size(group)
    sizes = shard_sizes(n_queries, world_size)
    longest = max(sizes)
    k = local_rows.shape[1]
    if local_rows.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local_rows does not match this rank's shard size")
    if local_rows.shape[0] == longest:
        padded = local_rows.contiguous()
    else:
        padded = torch.full((longest, k), -1, dtype=local_rows.dtype, device=local_rows.device)
        padded[:local_rows.shape[0]] = local_rows
    gathered = torch.empty((world_size * longest, k), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    if all(size == longest for size in sizes):
        return gathered
    pieces = [gathered[r * longest:r * longest + sizes[r]] for r in range(world_size)]
    return torch.cat(pieces, dim=0)


def slice_queries(q_rowptr, q_cols, q_maxint, begin, end):
    """The CSR slice of queries [begin, end) (host arrays)."""
    q_rowptr = np.asarray(q_rowptr)
    first, last = int(q_rowptr[begin]), int(q_rowptr[end])
    return (q_rowptr[begin:end + 1] - first).astype(np.int64), np.asarray(q_cols)[first:last], \
        np.asarray(q_maxint)[begin:end]
