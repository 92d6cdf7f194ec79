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


def test_shard_bounds_cover_exactly():
    import shard

    for n in (0, 1, 7, 8, 16384, 16385):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d


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
