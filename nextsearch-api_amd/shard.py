"""Query-parallel multi-GPU plumbing: the index is replicated on every GPU, each
rank scores its own contiguous shard of the query batch, and the fixed-size result blocks are
gathered with ONE collective per batch (RCCL all-gather over xGMI when the backend is "nccl";
gloo on CPU in the tests).  No collective touches the posting data path."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n queries for `rank`: ceil(n / world) per rank, last ranks may be short."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def gather_results(hits, nhits, found, out=None, group=None):
    """All-gather per-rank result blocks of EQUAL shape: hits [Q,K,3] int32 (score bits, seg, doc),
    nhits [Q] int32, found [Q] int64 -> ([W*Q,K,3], [W*Q], [W*Q]) on every rank, rank-major."""
    world = dist.get_world_size(group)
    if out is None:
        out = (torch.empty((world * hits.shape[0],) + tuple(hits.shape[1:]), dtype=hits.dtype, device=hits.device),
               torch.empty(world * nhits.shape[0], dtype=nhits.dtype, device=nhits.device),
               torch.empty(world * found.shape[0], dtype=found.dtype, device=found.device))
    dist.all_gather_into_tensor(out[0], hits, group=group)
    dist.all_gather_into_tensor(out[1], nhits, group=group)
    dist.all_gather_into_tensor(out[2], found, group=group)
    return out
