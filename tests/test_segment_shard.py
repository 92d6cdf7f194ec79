"""Segment-sharded multi-GPU path (SURVEY 8(e) alternative; nextsearch-api_amd/shard.py): every rank scores all
queries over ITS segments, one all-gather of the per-rank rows, one join.
  * CPU, world_size 2 on gloo: sub-index manifests, the exchange layout and the join's semantics, with the
    oracle standing in for each rank's scoring and a numpy join as the checker (there is no CPU join in the product);
  * GPU, one process: the ranks' engines run one after the other on the same device, their rows are stacked
    rank-major as the all-gather would deliver them, and ns_merge_rank_rows must reproduce the unsharded
    engine's results bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _rows_to_tensors(res, K):
    hits, nhits, found, _ = res
    t = np.zeros((len(nhits), K, 3), dtype=np.int32)
    t[:, :, 0] = hits["score"].view(np.int32)
    t[:, :, 1] = hits["seg"].view(np.int32)
    t[:, :, 2] = hits["doc"].view(np.int32)
    return torch.from_numpy(t), torch.from_numpy(nhits.astype(np.int32)), torch.from_numpy(found.astype(np.int64))


def _np_join(g_hits, g_nhits, g_found, seg_map, K):
    """Checker: the global heap over the ranks' rows — score desc, GLOBAL seg asc, doc asc."""
    W, Q = g_nhits.shape
    out = []
    for q in range(Q):
        cand = []
        for r in range(W):
            for i in range(int(g_nhits[r, q])):
                bits, seg, doc = (int(x) & 0xFFFFFFFF for x in g_hits[r, q, i])
                score = np.array([bits], dtype=np.uint32).view(np.float32)[0]
                cand.append((-float(score), int(seg_map[r, seg]), doc, bits))
        cand.sort()
        out.append(([(c[3], c[1], c[2]) for c in cand[:K]], int(g_found[:, q].sum())))
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, index_dir, shard_root, queries, K, ret):
    sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = os.path.join(shard_root, f"rank{rank}")
        mine = shard.write_segment_shard(index_dir, sd, rank, world)
        n_seg = len(shard.read_manifest(index_dir))
        seg_map = shard.seg_map_table(n_seg, world)
        assert seg_map[rank, :len(mine)].tolist() == mine
        ora = orc.Oracle(sd)                       # stands in for this rank's device scoring
        local = _rows_to_tensors(ora.search_batch(queries, K), K)
        ora.close()
        g_hits, g_nhits, g_found = shard.exchange_rank_rows(*local)
        assert g_hits.shape == (world, len(queries), K, 3)
        joined = _np_join(g_hits.numpy(), g_nhits.numpy(), g_found.numpy(), seg_map.numpy(), K)
        full = orc.Oracle(index_dir)
        fh, fn, ff, fu = full.search_batch(queries, K)
        full.close()
        ok = True
        for q in range(len(queries)):
            want = [(int(fh[q, i]["score"].view(np.uint32)), int(fh[q, i]["seg"]), int(fh[q, i]["doc"])) for i in range(int(fn[q]))]
            got, found = joined[q]
            ok = ok and got == want and found == int(ff[q])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_segment_shards_gloo(index_factory, tmp_path):
    import orc  # noqa: F401  (fails early if the oracle library is missing)

    d, _ = index_factory(5, 3000, 4096, 77, False)      # 5 segments over 2 ranks: 3 + 2
    import workloads
    queries = workloads.cfg4_queries(48)[:48] + ["", "zzzzunknownzzzz"]
    world, K = 2, 10
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, d, str(tmp_path), queries, K, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(280)
            assert p.exitcode == 0
        assert dict(ret) == {0: True, 1: True}


def test_owned_segments_partition():
    import shard

    for n in (0, 1, 5, 8, 9):
        for w in (1, 2, 3, 8):
            seen = sorted(i for r in range(w) for i in shard.owned_segments(n, r, w))
            assert seen == list(range(n))
            t = shard.seg_map_table(n, w)
            for r in range(w):
                ids = shard.owned_segments(n, r, w)
                assert t[r, :len(ids)].tolist() == ids and (t[r, len(ids):] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_segment_sharded_join_equals_unsharded_engine(world, golden_index, tmp_path):
    import nsbind
    import shard

    g, d, _ = golden_index("multi8")
    queries = g["queries"]
    full = nsbind.Engine(d, 0)
    try:
        n_seg = full.num_segments
        seg_map = shard.seg_map_table(n_seg, world)
        for K, flags in ((10, 0), (100, 0), (10, nsbind.NS_FLAG_AND)):
            rows = []
            for r in range(world):
                sd = str(tmp_path / f"w{world}_r{r}")
                shard.write_segment_shard(d, sd, r, world)
                e = nsbind.Engine(sd, 0)
                try:
                    rows.append(_rows_to_tensors(e.search_batch(queries, K, flags), K))
                finally:
                    e.close()
            g_hits = torch.stack([x[0] for x in rows]).cuda()
            g_nhits = torch.stack([x[1] for x in rows]).cuda()
            g_found = torch.stack([x[2] for x in rows]).cuda()
            nsbind.hip_lib().ns_ctx_set_stream(full.ctx, torch.cuda.current_stream().cuda_stream)
            oh, on, of = shard.merge_rank_rows(full.ctx, g_hits, g_nhits, g_found, seg_map, K)
            torch.cuda.synchronize()
            fh, fn, ff, fu = full.search_batch(queries, K, flags)
            assert np.array_equal(on.cpu().numpy().astype(np.uint32), fn)
            assert np.array_equal(of.cpu().numpy().astype(np.uint64), ff)
            got = oh.cpu().numpy()
            for q in range(len(queries)):
                n = int(fn[q])
                assert np.array_equal(got[q, :n, 0].view(np.uint32), fh[q, :n]["score"].view(np.uint32)), (world, K, q)
                assert np.array_equal(got[q, :n, 1].view(np.uint32), fh[q, :n]["seg"]) and np.array_equal(got[q, :n, 2].view(np.uint32), fh[q, :n]["doc"])
                assert np.all(got[q, n:, 2].view(np.uint32) == 0xFFFFFFFF)
    finally:
        full.close()
