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


# ------------------------------------------------------------------------------------------------
# Segment-sharded alternative: for an index that outgrows one GPU's HBM.  Rank r holds the
# segments i of the manifest with i % world == r and scores ALL queries over them; ONE all-gather of the
# fixed-size per-rank rows, then every rank joins them (ns_merge_rank_rows: the global heap of
# src/api_engine.cpp:434-435,485-492 — all segments compete on raw scores; found = sum over segments).
import os
import struct


def read_manifest(index_dir):
    with open(os.path.join(index_dir, "manifest.bin"), "rb") as f:
        b = f.read()
    (n,) = struct.unpack_from("<I", b, 0)
    pos, names = 4, []
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", b, pos)
        names.append(b[pos + 4: pos + 4 + ln])
        pos += 4 + ln
    return names


def owned_segments(n_segments, rank, world):
    """Global ids (positions in the manifest) of the segments `rank` holds: round-robin."""
    return list(range(rank, n_segments, world))


def write_segment_shard(index_dir, shard_dir, rank, world):
    """A sub-index for `rank`: a manifest naming only its segments, the segment directories shared through a
    symlink.  An Engine opened on it numbers its segments 0..; the returned list maps those local ids to the
    segments' positions in the full manifest (what the reference's hits carry, and the tie-break order)."""
    names = read_manifest(index_dir)
    mine = owned_segments(len(names), rank, world)
    os.makedirs(shard_dir, exist_ok=True)
    with open(os.path.join(shard_dir, "manifest.bin"), "wb") as f:
        f.write(struct.pack("<I", len(mine)) + b"".join(struct.pack("<I", len(names[i])) + names[i] for i in mine))
    link = os.path.join(shard_dir, "segments")
    if not os.path.lexists(link):
        os.symlink(os.path.abspath(os.path.join(index_dir, "segments")), link)
    return mine


def seg_map_table(n_segments, world):
    """[world, stride] int32: row r = global ids of rank r's local segments (padded with -1)."""
    stride = max(1, (n_segments + world - 1) // world)
    t = torch.full((world, stride), -1, dtype=torch.int32)
    for r in range(world):
        ids = owned_segments(n_segments, r, world)
        if ids:
            t[r, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return t


def exchange_rank_rows(hits, nhits, found, group=None):
    """All-gather of per-rank rows over the SAME queries: hits [Q,K,3] int32, nhits [Q] int32, found [Q] int64 ->
    rank-major ([W,Q,K,3], [W,Q], [W,Q]) on every rank."""
    world = dist.get_world_size(group)
    g = gather_results(hits, nhits, found, group=group)
    return g[0].view((world,) + tuple(hits.shape)), g[1].view(world, -1), g[2].view(world, -1)


def merge_rank_rows(ctx, g_hits, g_nhits, g_found, seg_map, k):
    """Join all-gathered rows on the device (k_merge_ranks behind ns_merge_rank_rows).  Device tensors in and out;
    fails loudly without the HIP library (there is no CPU merge in the product)."""
    import nsbind
    world, Q = g_nhits.shape
    out = (torch.empty((Q, k, 3), dtype=torch.int32, device=g_hits.device),
           torch.empty(Q, dtype=torch.int32, device=g_hits.device),
           torch.empty(Q, dtype=torch.int64, device=g_hits.device))
    sm = seg_map.to(device=g_hits.device, dtype=torch.int32).contiguous() if seg_map is not None else None
    rc = nsbind.hip_lib().ns_merge_rank_rows(ctx, g_hits.data_ptr(), g_nhits.data_ptr(), g_found.data_ptr(), world, Q, k,
                                             sm.data_ptr() if sm is not None else None, sm.shape[1] if sm is not None else 0,
                                             out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    if rc != 0:
        raise RuntimeError("ns_merge_rank_rows: " + nsbind.hip_lib().ns_last_error(ctx).decode())
    return out
