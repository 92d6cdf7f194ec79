#!/usr/bin/env python3
"""Diagnostic: where the time of bench.py's N > 1 step goes (2 ranks on one GPU over gloo)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import numpy as np, torch, torch.distributed as dist
import nsbind, shard, workloads
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(backend=sys.argv[1] if len(sys.argv) > 1 else "gloo")
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
os.environ["NS_RELOAD_WARMUP"] = "0"
eng = nsbind.Engine(idx, 0)
Q, K = 16384, 10
qs = workloads.cfg5_queries(Q)
lo, hi = shard.shard_bounds(Q, rank, world); per = (Q + world - 1) // world
qd, refs, _ = eng.build_refs(qs[lo:hi])
nbytes, off_n, off_f = shard.packed_layout(per, K)
D = 3
blocks = [shard.alloc_packed(per, K, "cuda") for _ in range(D)]
gathered = [torch.empty(world * nbytes, dtype=torch.uint8, device="cuda") for _ in range(D)]
torch.cuda.synchronize()
for mode in ("sync_default_stream", "async_ext_stream"):
    acc = [0.0] * 5
    n = 12
    for i in range(n):
        t0 = time.perf_counter()
        b = nsbind.prepare_raw(eng.ctx, qd, refs, K)
        t1 = time.perf_counter()
        blk = blocks[i % D]
        b.bind_outputs(blk.data_ptr(), blk.data_ptr() + off_n, blk.data_ptr() + off_f)
        b.run()
        t2 = time.perf_counter()
        if mode == "sync_default_stream":
            b.sync()
            t3 = time.perf_counter()
            shard.gather_packed(blk, gathered[i % D])
            t4 = time.perf_counter()
        else:
            st = torch.cuda.ExternalStream(b.stream)
            with torch.cuda.stream(st):
                work = shard.gather_packed(blk, gathered[i % D], async_op=True)
                t3 = time.perf_counter()
                work.wait()
                ev = torch.cuda.Event(); ev.record(st)
            ev.synchronize()
            t4 = time.perf_counter()
        b.close()
        t5 = time.perf_counter()
        if i >= 2:
            for j, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[j] += d
    if rank == 0:
        print(mode, "ms per step: prepare %.2f | bind+run %.2f | sync-or-issue %.2f | gather/wait %.2f | close %.2f" % tuple(1e3 * a / (n - 2) for a in acc), flush=True)
dist.barrier(); eng.close(); dist.destroy_process_group()
