"""Query-parallel multi-GPU plumbing: the index is replicated on every GPU, each
rank scores its own contiguous shard of the query batch, and the fixed-size result blocks are
gathered with ONE collective per batch — hits, nhits and found travel as one packed block (RCCL all-gather over xGMI
when the backend is "nccl"; gloo on CPU in the tests).  No collective touches the posting data path."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n queries for `rank`: ceil(n / world) per rank, last ranks may be short."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _align(n, a=256):
    return (n + a - 1) // a * a


def packed_layout(rows, k):
    """One result block = [hits rows*k*12 B | nhits rows*4 B | found rows*8 B], sections 256-byte aligned, so that a
    rank's whole answer is ONE buffer and the step's exchange ONE collective.  -> (nbytes, off_nhits, off_found)."""
    off_n = _align(rows * k * 12)
    off_f = off_n + _align(rows * 4)
    return off_f + _align(rows * 8), off_n, off_f


def alloc_packed(rows, k, device, world=1):
    nbytes, _, _ = packed_layout(rows, k)
    return torch.zeros(world * nbytes, dtype=torch.uint8, device=device)


def packed_views(buf, rows, k, world=1):
    """Typed views of `world` result blocks laid out back to back in `buf` (uint8): hits [W, rows, k, 3] int32
    (score bits, seg, doc), nhits [W, rows] int32, found [W, rows] int64.  No copies."""
    nbytes, off_n, off_f = packed_layout(rows, k)
    blocks = buf.view(world, nbytes)
    hits = blocks[:, : rows * k * 12].view(torch.int32).view(world, rows, k, 3)
    nhits = blocks[:, off_n: off_n + rows * 4].view(torch.int32).view(world, rows)
    found = blocks[:, off_f: off_f + rows * 8].view(torch.int64).view(world, rows)
    return hits, nhits, found


def gather_packed(block, out, group=None, async_op=False):
    """THE exchange step of the query-sharded path: every rank contributes its packed result block (equal size on all
    ranks: short last shards are padded to ceil(Q / world) rows), every rank receives all of them, rank-major."""
    return dist.all_gather_into_tensor(out, block, group=group, async_op=async_op)


def unshard(gathered, n_queries, world, k):
    """The global batch's results from the gathered blocks: drops the padding rows of short shards.
    -> hits [Q, k, 3] int32, nhits [Q] int32, found [Q] int64 (copies)."""
    per = (n_queries + world - 1) // world
    hits, nhits, found = packed_views(gathered, per, k, world)
    keep = [min(per, max(0, n_queries - r * per)) for r in range(world)]
    return (torch.cat([hits[r, : keep[r]] for r in range(world)]), torch.cat([nhits[r, : keep[r]] for r in range(world)]),
            torch.cat([found[r, : keep[r]] for r in range(world)]))


def gather_results(hits, nhits, found, out=None, group=None):
    """All-gather per-rank result arrays of EQUAL shape: hits [Q,K,3] int32 (score bits, seg, doc), nhits [Q] int32,
    found [Q] int64 -> ([W*Q,K,3], [W*Q], [W*Q]) on every rank, rank-major.  The three arrays travel as ONE packed
    block in ONE collective (packed_layout)."""
    world = dist.get_world_size(group)
    Q, K = hits.shape[0], hits.shape[1]
    block = alloc_packed(Q, K, hits.device)
    bh, bn, bf = packed_views(block, Q, K)
    bh[0].copy_(hits); bn[0].copy_(nhits); bf[0].copy_(found)
    gathered = torch.empty(world * block.numel(), dtype=torch.uint8, device=hits.device)
    gather_packed(block, gathered, group=group)
    gh, gn, gf = packed_views(gathered, Q, K, world)
    res = (gh.reshape(world * Q, K, 3), gn.reshape(world * Q), gf.reshape(world * Q))
    if out is not None:
        out[0].copy_(res[0]); out[1].copy_(res[1]); out[2].copy_(res[2])
        return out
    return tuple(t.contiguous() for t in res)


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
