"""world_size-2 `gloo` test of the N>1 path on CPU: contiguous query shards + one all-gather of the
fixed-size result blocks (nextsearch-api_amd/shard.py, used by bench.py with the nccl==RCCL backend).
No scoring happens here (there is no CPU scoring path): each rank fabricates the result block of
its own shard from a known global table, and every rank must end up with the whole table."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Q, K, ret):
    sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
    import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        table_hits = torch.randint(0, 2**31 - 1, (world * Q, K, 3), dtype=torch.int32, generator=g)
        table_n = torch.randint(0, K + 1, (world * Q,), dtype=torch.int32, generator=g)
        table_f = torch.randint(0, 10**9, (world * Q,), dtype=torch.int64, generator=g)
        lo, hi = shard.shard_bounds(world * Q, rank, world)
        assert (lo, hi) == (rank * Q, (rank + 1) * Q)
        out = shard.gather_results(table_hits[lo:hi].contiguous(), table_n[lo:hi].contiguous(), table_f[lo:hi].contiguous())
        ok = torch.equal(out[0], table_hits) and torch.equal(out[1], table_n) and torch.equal(out[2], table_f)
        # MAX-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _strong_worker(rank, world, port, Q, K, ret):
    """bench.py --scaling strong: ONE global batch of Q queries cut into contiguous shards of ceil(Q / world) (the last
    one short or empty), every rank fills its packed block for its own shard only, ONE all-gather (here asynchronous,
    as bench.py issues it), and every rank reassembles the whole batch, padding rows dropped."""
    sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
    import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(99)
        table_hits = torch.randint(-2**31, 2**31 - 1, (Q, K, 3), dtype=torch.int32, generator=g)
        table_n = torch.randint(0, K + 1, (Q,), dtype=torch.int32, generator=g)
        table_f = torch.randint(0, 2**40, (Q,), dtype=torch.int64, generator=g)
        per = (Q + world - 1) // world
        lo, hi = shard.shard_bounds(Q, rank, world)
        block = shard.alloc_packed(per, K, "cpu")
        bh, bn, bf = shard.packed_views(block, per, K)
        n = hi - lo
        bh[0, :n].copy_(table_hits[lo:hi]); bn[0, :n].copy_(table_n[lo:hi]); bf[0, :n].copy_(table_f[lo:hi])
        bh[0, n:] = -1                                     # padding rows of a short shard: must never reach the result
        gathered = torch.empty(world * block.numel(), dtype=torch.uint8)
        work = shard.gather_packed(block, gathered, async_op=True)
        work.wait()
        hits, nhits, found = shard.unshard(gathered, Q, world, K)
        ok = hits.shape[0] == Q and torch.equal(hits, table_hits) and torch.equal(nhits, table_n) and torch.equal(found, table_f)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_exactly():
    import shard

    for n in (0, 1, 7, 8, 16384, 16385):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,Q", [(2, 101), (3, 100), (2, 1)])
def test_strong_sharding_uneven_shards_one_collective_gloo(world, Q):
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        procs = [ctx.Process(target=_strong_worker, args=(r, world, port, Q, 10, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(200)
            assert p.exitcode == 0
        assert dict(ret) == {r: True for r in range(world)}


def test_packed_layout_is_what_the_c_abi_binds():
    """bench.py binds a batch's outputs (ns_batch_bind_outputs) to the three sections of ONE packed block."""
    import shard
    for rows, k in ((2048, 10), (1, 1), (4096, 100), (513, 33)):
        nbytes, off_n, off_f = shard.packed_layout(rows, k)
        assert off_n % 256 == 0 and off_f % 256 == 0 and nbytes % 256 == 0
        assert off_n >= rows * k * 12 and off_f >= off_n + rows * 4 and nbytes >= off_f + rows * 8
        buf = shard.alloc_packed(rows, k, "cpu", world=3)
        h, n, f = shard.packed_views(buf, rows, k, world=3)
        assert h.shape == (3, rows, k, 3) and n.shape == (3, rows) and f.shape == (3, rows)
        h[1, rows - 1, k - 1, 2] = 7; n[2, 0] = 5; f[0, rows - 1] = 9
        assert buf.view(3, nbytes)[1, rows * k * 12 - 4] == 7 and buf.view(3, nbytes)[2, off_n] == 5 and buf.view(3, nbytes)[0, off_f + (rows - 1) * 8] == 9


@pytest.mark.timeout(180)
def test_two_rank_gather_gloo():
    world, Q, K = 2, 64, 10
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, Q, K, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(150)
            assert p.exitcode == 0
        assert dict(ret) == {0: True, 1: True}
