#!/usr/bin/env python3
"""bench.py — queries/sec of the MI355X BM25 hot path, timed at the C-ABI boundary, + fraction of the HBM roofline.

A "step" is one pass of the hot path over one batch of synthetic queries: the batch's term refs sit in HOST memory in
the C-ABI's layout (what the host facade hands down: lexicon probes and idf done), the index is resident in HBM, and
the step runs ns_batch_prepare (regroup, cut into work items, upload of the descriptors) -> scoring + merge kernels ->
results back in host memory (N = 1), resp. -> ONE RCCL all-gather of the packed result blocks (N > 1).  Steps are
pipelined on the one ctx (prepare(i+1) || run(i) || fetch(i-1), include/nextsearch_hip.h: NS_RUN_FETCH,
ns_ctx_set_overlap), as a serving loop would run them.  SURVEY.md §8(d): "QPS = batch wall time at the C-ABI boundary,
H2D of descriptors and D2H of results included, segment upload excluded".

Default workload = BASELINE config 5: Zipf-skewed 1-8 term mix, K = 10, OR, 1M-doc synthetic CORD-19-shaped index,
ONE global batch of 16384 queries.  N > 1: the index is replicated and the batch is cut into N contiguous shards of
ceil(16384 / N) queries (strong scaling, the way BASELINE configs 4 and 5 are written); `--scaling weak` gives every
rank a full 16384-query batch of its own instead.

    python bench.py [--gpus N --steps K --warmup W] [--config cfg5|cfg3|cfg4|cfg2] [--scaling strong|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Beside the contract's keys it carries
  roofline       the scoring kernel alone, descriptors resident: algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  kernel_only    the device-only rate of the same batch (what round 1 reported as `value`)
  hbm_resident   (N = 1) the same query law over a 20 x 1M-doc index — 1.1 GB of postings, beyond the 256 MiB Infinity
                 Cache — with its own roofline numbers
  cpu_baseline   (N = 1) the REAL reference engine on one host core of this box (+ cpu_baselines: the as-shipped
                 no-flag build, and the oracle port on all cores as the generous upper bound)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from collections import deque

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "nextsearch-api_amd")
sys.path.insert(0, PKG)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what streaming kernels reach

WORKLOAD_TEXT = {
    "cfg5": "BASELINE config 5 query law: 1-8 terms (1+Poisson(2)), 30% hot ranks [1,32] / 70% log-uniform tail, OR, k=10",
    "cfg3": "BASELINE config 3: 5-term disjunctive, rank~1/r on [1,5000], k=100",
    "cfg4": "BASELINE config 4: cfg3 law over 8 x 125k-doc segments, k=10",
    "cfg2": "BASELINE config 2: 2-term conjunctive (AND extension), ranks U[10,1000], k=10, 100k docs",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg5", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = ONE global batch cut into N contiguous shards (default, as BASELINE's configs are written); weak = a full batch per rank")
    ap.add_argument("--queries", type=int, default=0, help="queries in the global batch (0 = the config's batch)")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight per ctx in the pipelined loop")
    ap.add_argument("--no-overlap", action="store_true", help="keep every batch on one stream (default: batches alternate between two)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--min-items", type=int, default=0)
    ap.add_argument("--split", type=int, default=0, help="work units per work item (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the reference CPU baseline sample (0 = skip all CPU baselines)")
    ap.add_argument("--index-dir", default="", help="reuse/generate the index here instead of a temp dir")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI; the measured configuration) or gloo (rehearsal of the N > 1 code path with several ranks on ONE GPU)")
    ap.add_argument("--share", type=int, default=1, choices=[0, 1, 2],
                    help="ns_ctx_share_scores: 1 (the library's default) a batch that names its lists often enough computes each distinct list's "
                         "BM25 term scores once per run; 0 every posting is scored in place for every query that names it; 2 always share")
    ap.add_argument("--no-impact-leg", action="store_true", help="skip the extra measurement over the optional impact streams")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the extra measurement over the 20-segment (HBM-resident) index")
    ap.add_argument("--hbm-segments", type=int, default=20)
    ap.add_argument("--hbm-queries", type=int, default=2048)
    ap.add_argument("--hbm-only", action="store_true", help="profiling runs: ONLY the hbm_resident leg (prints its object as the JSON line)")
    ap.add_argument("--kernel-only", action="store_true", help="profiling runs: only the kernel leg (descriptors resident), no pipelined loop, no extra legs")
    return ap.parse_args()


def host_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def lib_info():
    """Which native libraries this process measured: path and sha256 (first 16 hex digits) of the two in-tree .so files."""
    import hashlib
    out = {}
    for name in ("libnextsearch_hip.so", "libnextsearch_host.so"):
        p = os.path.join(PKG, name)
        try:
            with open(p, "rb") as f:
                out[name] = hashlib.sha256(f.read()).hexdigest()[:16]
        except OSError:
            out[name] = None
    return out


def cpu_baselines(index_dir, queries, k, budget_s):
    """The CPU side of SURVEY 8(d), on THIS box's host cores, each on a bounded sample of the same workload:
      reference -O2, 1 core   cord19::Engine::search compiled from the reference's sources (oracle/_ref/ref_driver); one
                              core because the reference serialises every search behind Engine::mtx (src/api_engine.cpp:372)
      reference as shipped    the same with no optimisation flag, as its CMakeLists.txt builds it
      port, all cores         oracle/bm25_oracle.c (dense accumulator array, canonical tie order: faster than the
                              reference's hash map), one replica per core: the generous CPU upper bound
    Returns (headline dict, list of all)."""
    info = host_info()
    out = []
    with tempfile.TemporaryDirectory(prefix="ns_cpu_") as tmp:
        qpath, qpath_all = os.path.join(tmp, "q.txt"), os.path.join(tmp, "q_all.txt")
        with open(qpath, "w") as f:
            f.write("\n".join(queries[:4096]) + "\n")
        with open(qpath_all, "w") as f:                    # all cores need more work than one core: the whole batch, 8 times over
            for _ in range(8):
                f.write("\n".join(queries) + "\n")
        runs = [
            ("reference", os.path.join(ROOT, "oracle", "_ref", "ref_driver"), budget_s, 1,
             "cord19::Engine::search built -O2 from the reference sources, search cache emptied per call"),
            ("reference", os.path.join(ROOT, "oracle", "_ref", "ref_driver_O0"), min(budget_s, 6.0), 1,
             "the same engine as shipped: no optimisation flag (the reference's CMakeLists.txt sets none)"),
            ("port", os.path.join(ROOT, "oracle", "bm25_oracle_cli"), min(budget_s, 5.0), info["nproc"] or 1,
             "oracle/bm25_oracle.c (dense accumulators, canonical order), queries partitioned over all cores"),
        ]
        for kind, exe, secs, threads, what in runs:
            if not os.path.exists(exe):
                continue
            try:
                cmd = [exe, "time", index_dir, qpath_all if kind == "port" else qpath, str(k), str(secs)] + ([str(threads)] if kind == "port" else [])
                res = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=secs * 6 + 180).stdout
                r = json.loads(res.strip().splitlines()[-1])
                out.append({"value": r["qps"], "unit": "queries/s", "cores": threads, "kind": kind, "nproc": info["nproc"], "cpu_model": info["cpu_model"],
                            "sample": f"{r['queries']} queries of the same workload, {r['seconds']:.1f} s; {what}"})
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] cpu baseline via {exe} failed: {e}\n")
    if not out:
        return None, []
    return out[0], out


def run_hbm_leg(args, nsbind, np, gen, seed, docs, K, flags, device, traffic_db):
    """The same query law over an index that cannot sit in the 256 MiB Infinity Cache: S x 1M docs (every query scans all
    S segments: S x the postings per query; 20 segments = 1.1 GB of postings + 0.55 GB of per-posting norms).  Kernel leg
    only: descriptors resident, the scoring launch under HIP events."""
    S, Qb = args.hbm_segments, args.hbm_queries
    with tempfile.TemporaryDirectory(prefix="ns_bench_big_") as big:
        bidx = os.path.join(big, "index")
        nsbind.gen_index(bidx, S, docs, 65536, 1337, False)
        beng = nsbind.Engine(bidx, device)
        beng.set_tuning(args.variant, args.min_items, args.split)
        beng.share_scores(args.share)
        bq = gen(Qb, seed)
        bqd, brefs, _ = beng.build_refs(bq)
        bb = nsbind.prepare_raw(beng.ctx, bqd, brefs, K, flags)
        for _ in range(3):
            bb.run(timed=False)
        bb.sync()
        n_big = max(5, args.steps // 2)
        for _ in range(n_big):
            bb.run(timed=True)
        bb.sync()
        binf = bb.info()
        bh, bn, bf = bb.fetch()
        assert (bn == np.minimum(bf, K)).all()
        bb.close()
        bms = binf.sum_score_kernel_ms / max(binf.timed_runs, 1)
        bach = binf.algo_bytes / (bms * 1e-3) / 1e9
        dev_bytes = sum(beng.segment_info(s)["n_postings"] for s in range(S)) * 12
        tr = traffic_db.get(f"cfg5_big{S}_q{Qb}")
        leg = {
            "what": f"cfg5 query law, {Qb} queries over {S} segments x {docs} docs (every query scans every segment), kernel only",
            "index_bytes_on_device": int(dev_bytes), "postings_per_query": binf.postings / max(Qb, 1), "work_items": binf.n_items,
            "kernel_ms": bms, "queries_per_s_kernel": Qb / (bms * 1e-3), "timed_launches": int(binf.timed_runs),
            "term_scores": (f"shared inside the batch: {binf.shared_lists} distinct lists, {binf.shared_postings} postings scored once per launch by k_share_scores (inside kernel_ms)"
                            if binf.flags & nsbind.NS_INFO_SHARED else "in place"),
            "roofline": {"bound": "hbm", "achieved": bach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bach / HBM_PEAK_GBS,
                         "algo_bytes_per_launch": int(binf.algo_bytes), "traffic": tr,
                         "traffic_over_algo": (tr / binf.algo_bytes) if tr else None,
                         "traffic_source": "L2-miss bytes per launch from the builder's rocprofv3 --pmc FETCH_SIZE pass of `bench.py --hbm-only` (x2 gfx950 correction), profiles/r03/final_hbm_leg_pmc_fetch.csv; not measured in this run",
                         "limited_by": "instruction issue / memory latency, no longer bandwidth: with the XCD-aware launch order (items that share a list and a doc range meet in one L2) the launch moves ~1.3x its algorithmic bytes at ~4.6 TB/s of L2-miss traffic, below the ~6.3 TB/s this part streams at (round 2: 1.87x at 6.2 TB/s, saturated)"},
        }
        beng.close()
        return leg


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        sys.stderr.write(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE\n")
    n_gpus = max(world, 1)

    import numpy as np
    import torch

    import nsbind
    import shard
    import workloads

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (there is no CPU path); torch.cuda.is_available() is False")
    if args.backend == "gloo":   # rehearsal: ranks share the devices that exist
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if n_gpus > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    L = nsbind.hip_lib()

    gen, q_default, K, flags, (nseg, docs) = workloads.WORKLOADS[args.config]
    if args.hbm_only:
        os.environ["NS_RELOAD_WARMUP"] = "0"
        tdb = {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tdb = json.load(f)
        except Exception:   # noqa: BLE001
            pass
        leg = run_hbm_leg(args, nsbind, np, gen, {"cfg2": 2002, "cfg3": 2003, "cfg4": 2004, "cfg5": 2005}[args.config], docs, K, flags, local_rank, tdb)
        print(json.dumps({"hbm_resident": leg, "library": lib_info()}), flush=True)
        return
    Q = args.queries or q_default                       # the global batch (strong) / the per-rank batch (weak)
    seed = {"cfg2": 2002, "cfg3": 2003, "cfg4": 2004, "cfg5": 2005}[args.config]
    strong = args.scaling == "strong" or n_gpus == 1
    if strong:
        all_queries = gen(Q, seed)                      # ONE global batch, the same on every rank
        lo, hi = shard.shard_bounds(Q, rank, n_gpus)    # this rank's contiguous shard: ceil(Q / N) queries, the last may be short
        queries = all_queries[lo:hi]
        per = (Q + n_gpus - 1) // n_gpus                # rows of every rank's result block (short shards are padded)
        global_batch = Q
    else:
        queries = gen(Q, seed + 7919 * rank)            # a full batch of its own per rank
        per = Q
        global_batch = n_gpus * Q
    Qr = len(queries)

    tmp = None
    if args.index_dir:
        index_dir = args.index_dir
        if rank == 0 and not os.path.exists(os.path.join(index_dir, "manifest.bin")):
            nsbind.gen_index(index_dir, nseg, docs, 65536, 1337, False)
        if dist is not None:
            dist.barrier()
    else:
        tmp = tempfile.TemporaryDirectory(prefix=f"ns_bench_r{rank}_")
        index_dir = os.path.join(tmp.name, "index")
        nsbind.gen_index(index_dir, nseg, docs, 65536, 1337, False)

    os.environ["NS_RELOAD_WARMUP"] = "0"   # no warm-up query at reload: rocprof / PMC summaries of this command then hold the batch launches only
    eng = nsbind.Engine(index_dir, local_rank)
    eng.set_tuning(args.variant, args.min_items, args.split)
    eng.share_scores(args.share)
    # the step's INPUT: the shard's term refs in host memory, in the C-ABI's layout (tokenise + lexicon probes + idf done)
    qd, refs, usable = eng.build_refs(queries)
    assert usable.all()

    # ---- kernel leg: descriptors resident in HBM, the scoring launch alone under HIP events (the roofline's numbers) ----
    kb = nsbind.prepare_raw(eng.ctx, qd, refs, K, flags)
    # the first launches after an idle period run ~10 % slow while the device clocks ramp (profiles/r01/final_cfg5_kernel_trace_head.csv)
    for _ in range(6):
        kb.run(timed=False)
    kb.sync()
    for _ in range(args.warmup):
        kb.run(timed=False)
    kb.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kb.run(timed=True)
    kb.sync()   # also reads the HIP events recorded on the stream
    kernel_elapsed = time.perf_counter() - t0
    kinfo = kb.info()
    k_hits, k_nhits, k_found = kb.fetch()
    assert (k_nhits == np.minimum(k_found, K)).all(), "result sanity check failed"
    kb.close()
    score_ms = kinfo.sum_score_kernel_ms / max(kinfo.timed_runs, 1)
    total_ms = kinfo.sum_total_ms / max(kinfo.timed_runs, 1)
    shared = bool(kinfo.flags & nsbind.NS_INFO_SHARED)
    # the same batch with sharing off (every posting scored in place, once per query that names it): reported next to the roofline
    in_place_leg = None
    if shared and not args.kernel_only:
        eng.share_scores(0)
        pb = nsbind.prepare_raw(eng.ctx, qd, refs, K, flags)
        for _ in range(max(args.warmup, 2)):
            pb.run(timed=False)
        pb.sync()
        for _ in range(args.steps):
            pb.run(timed=True)
        pb.sync()
        pinf = pb.info()
        p_hits, p_nhits, p_found = pb.fetch()
        pb.close()
        eng.share_scores(args.share)
        assert not (pinf.flags & (nsbind.NS_INFO_SHARED | nsbind.NS_INFO_IMPACTS))
        assert p_hits.tobytes() == k_hits.tobytes() and p_nhits.tobytes() == k_nhits.tobytes() and p_found.tobytes() == k_found.tobytes(), \
            "shared term scores changed the results"
        pms = pinf.sum_score_kernel_ms / max(pinf.timed_runs, 1)
        in_place_leg = {"what": "same batch, kernel only, ns_ctx_share_scores(0): the BM25 term score of a posting evaluated once per query that names its list",
                        "kernel_ms": pms, "achieved": pinf.algo_bytes / (pms * 1e-3) / 1e9 if pms > 0 else 0.0,
                        "frac": (pinf.algo_bytes / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pms > 0 else 0.0, "identical_results": True}

    # ---- value leg: K pipelined steps at the C-ABI boundary ----
    # A serving loop sees a different batch every step: the steps rotate over ROT distinct batches of the same query law and
    # size (batch 0 = the kernel leg's; batch i = seed + 104729 i), so the device never sees the same launch twice in a row
    # (other items, other launch order, other L2 / Infinity-Cache footprint).  Every batch's expected answer is computed once,
    # untimed, one batch at a time; the timed region's last step is compared with its batch's.
    ROT = 4
    elapsed = None
    if not args.kernel_only:
        if strong:
            rot_queries = [all_queries] + [gen(Q, seed + 104729 * i) for i in range(1, ROT)]
            rot_local = [qs_[lo:hi] for qs_ in rot_queries]
        else:
            rot_queries = None
            rot_local = [queries] + [gen(Q, seed + 7919 * rank + 104729 * i) for i in range(1, ROT)]
        rot = []          # per batch: (qd, refs) of THIS rank's shard
        rot_expect = []   # per batch: (hits, nhits, found) of this rank's shard, from an unpipelined run
        for i, qs_ in enumerate(rot_local):
            if i == 0:
                rot.append((qd, refs)); rot_expect.append((k_hits, k_nhits, k_found))
                continue
            qd_i, refs_i, us_i = eng.build_refs(qs_)
            assert us_i.all() and len(qd_i) == Qr
            rb = nsbind.prepare_raw(eng.ctx, qd_i, refs_i, K, flags)
            rb.run(timed=False); rb.sync()
            rot.append((qd_i, refs_i)); rot_expect.append(rb.fetch())
            rb.close()
        L.ns_ctx_set_overlap(eng.ctx, 0 if args.no_overlap else 1)
        depth = max(1, args.depth)
        if dist is None:
            out = [(np.empty((Qr, K), dtype=nsbind.HIT_DTYPE), np.empty(Qr, np.uint32), np.empty(Qr, np.uint64)) for _ in range(2)]

            def run_steps(n):
                last = None
                for res in nsbind.pipelined_search(eng.ctx, [rot[i % ROT] for i in range(n)], K, flags, out=out, depth=depth):
                    last = res
                return last

            run_steps(args.warmup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = run_steps(args.steps)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            # the timed region's results are real: the last step's host buffers equal what its batch gives one batch at a time
            e_hits, e_nhits, e_found = rot_expect[(args.steps - 1) % ROT]
            assert last[0].tobytes() == e_hits.tobytes() and last[1].tobytes() == e_nhits.tobytes() and last[2].tobytes() == e_found.tobytes(), \
                "pipelined results differ from the unpipelined run of the same batch"
        else:
            # every rank: prepare(own shard) -> kernels into ITS packed block -> ONE all-gather of the blocks; up to `depth`
            # steps in flight (the host only waits for the step that left the pipeline)
            nbytes, off_n, off_f = shard.packed_layout(per, K)
            blocks = [shard.alloc_packed(per, K, "cuda") for _ in range(depth)]
            gathered = [torch.empty(n_gpus * nbytes, dtype=torch.uint8, device="cuda") for _ in range(depth)]
            torch.cuda.synchronize()

            # batches alternate between two torch streams (the ctx works on whichever it is given): the collective of
            # step i — enqueued behind stream i's kernels, as any torch collective is ordered after the current stream —
            # then overlaps the kernels of step i+1 on the other stream
            streams = [torch.cuda.Stream() for _ in range(1 if args.no_overlap else 2)]

            def run_steps(n):
                flight = deque()
                for i in range(n):
                    if len(flight) >= depth:
                        flight.popleft()()
                    st = streams[i % len(streams)]
                    L.ns_ctx_set_stream(eng.ctx, st.cuda_stream)
                    b = nsbind.prepare_raw(eng.ctx, *rot[i % ROT], K, flags)
                    blk = blocks[i % depth]
                    b.bind_outputs(blk.data_ptr(), blk.data_ptr() + off_n, blk.data_ptr() + off_f)
                    b.run(timed=False)
                    with torch.cuda.stream(st):
                        shard.gather_packed(blk, gathered[i % depth])   # ONE collective per step
                        ev = torch.cuda.Event()
                        ev.record(st)

                    def retire(b=b, ev=ev):
                        ev.synchronize()
                        b.close()
                    flight.append(retire)
                while flight:
                    flight.popleft()()
                L.ns_ctx_set_stream(eng.ctx, None)

            run_steps(args.warmup)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            dist.barrier()
            elapsed = time.perf_counter() - t0
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # The gathered block of the LAST timed step, checked on every rank: this rank's own rows — hits, nhits, found — equal
            # what its shard of that batch gives one batch at a time; and under strong scaling (every rank knows the whole
            # global batch) ALL ranks' rows equal this rank's own unsharded answer to the global batch, computed here once.
            gh, gn, gf = shard.packed_views(gathered[(args.steps - 1) % depth], per, K, n_gpus)
            li = (args.steps - 1) % ROT
            e_hits, e_nhits, e_found = rot_expect[li]
            g_hits = gh.cpu().numpy().view(nsbind.HIT_DTYPE).reshape(n_gpus, per, K)
            g_nhits = gn.cpu().numpy().astype(np.uint32)
            g_found = gf.cpu().numpy().astype(np.uint64)
            assert g_hits[rank, :Qr].tobytes() == e_hits.tobytes() and np.array_equal(g_nhits[rank, :Qr], e_nhits) and np.array_equal(g_found[rank, :Qr], e_found), \
                "gathered results (own rows) differ from the unpipelined run of the same batch"
            if strong:
                gq, grefs, gus = eng.build_refs(rot_queries[li])
                assert gus.all()
                gb = nsbind.prepare_raw(eng.ctx, gq, grefs, K, flags)
                gb.run(timed=False); gb.sync()
                w_hits, w_nhits, w_found = gb.fetch()
                gb.close()
                for r in range(n_gpus):
                    a, b_ = shard.shard_bounds(Q, r, n_gpus)
                    assert g_hits[r, :b_ - a].tobytes() == w_hits[a:b_].tobytes() and np.array_equal(g_nhits[r, :b_ - a], w_nhits[a:b_]) \
                        and np.array_equal(g_found[r, :b_ - a], w_found[a:b_]), f"gathered rows of rank {r} differ from the unsharded answer"
        L.ns_ctx_set_overlap(eng.ctx, 0)

    # ---- extra legs (N = 1 only; reported next to the headline, never as `value`) ----
    impact_leg = None
    if n_gpus == 1 and args.variant == 0 and not args.no_impact_leg and not args.kernel_only:
        t0 = time.perf_counter()
        eng.build_impacts()
        build_s = time.perf_counter() - t0
        b2 = nsbind.prepare_raw(eng.ctx, qd, refs, K, flags)
        for _ in range(args.warmup):
            b2.run(timed=False)
        b2.sync()
        for _ in range(args.steps):
            b2.run(timed=True)
        b2.sync()
        inf2 = b2.info()
        i_hits, i_nhits, i_found = b2.fetch()
        same = i_hits.tobytes() == k_hits.tobytes() and i_nhits.tobytes() == k_nhits.tobytes() and i_found.tobytes() == k_found.tobytes()
        assert inf2.flags & nsbind.NS_INFO_IMPACTS, "the impact leg did not read the impact streams"
        assert same, "impact-stream results differ from the {docId, tf} path"
        k2 = inf2.sum_score_kernel_ms / max(inf2.timed_runs, 1)
        impact_leg = {
            "what": "same batch, kernel only, postings read as {docId, precomputed fp32 term score} (optional second stream, built once per list)",
            "kernel_ms": k2, "achieved": inf2.algo_bytes / (k2 * 1e-3) / 1e9 if k2 > 0 else 0.0,
            "frac": (inf2.algo_bytes / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS) if k2 > 0 else 0.0,
            "build_s": build_s, "identical_results": same,
        }
        b2.close()
        eng.use_impacts(False)

    # block-max pruning: the same batch with the single-term queries skipping the blocks that cannot enter
    # their top-K.  Reported next to the headline, never as `value` or `roofline`: those stay on the exhaustive path, where
    # "algorithmic bytes" are bytes actually scanned.
    pruned_leg = None
    st_queries = [q for q in queries if len(q.split()) == 1]   # (the workloads' words are all index terms: one word = one scored term)
    if n_gpus == 1 and args.variant == 0 and not args.no_impact_leg and not args.kernel_only and flags == 0 and st_queries:
        t0 = time.perf_counter()
        eng.build_blockmax()
        build_s = time.perf_counter() - t0
        eng.use_pruning(True)
        b3 = nsbind.prepare_raw(eng.ctx, qd, refs, K, flags)
        for _ in range(args.warmup):
            b3.run(timed=False)
        b3.sync()
        for _ in range(args.steps):
            b3.run(timed=True)
        b3.sync()
        inf3 = b3.info()
        p_hits, p_nhits, p_found = b3.fetch()
        same = p_hits.tobytes() == k_hits.tobytes() and p_nhits.tobytes() == k_nhits.tobytes() and p_found.tobytes() == k_found.tobytes()
        assert inf3.flags & nsbind.NS_INFO_PRUNED, "the pruned leg did not take the block-max body"
        assert same, "pruned results differ from the exhaustive path"
        k3 = inf3.sum_score_kernel_ms / max(inf3.timed_runs, 1)
        single = len(st_queries)
        # the same two ways for the batch's single-term queries ALONE (in the full batch they are not what finishes last, so
        # the batch's time hardly moves; a batch of them shows what the skipped blocks are worth)
        subset = None
        if len(st_queries) >= 64:
            sqd, srefs, _ = eng.build_refs(st_queries)
            times = {}
            for mode in (False, True):
                eng.use_pruning(mode)
                sb = nsbind.prepare_raw(eng.ctx, sqd, srefs, K, flags)
                for _ in range(args.warmup):
                    sb.run(timed=False)
                sb.sync()
                for _ in range(args.steps):
                    sb.run(timed=True)
                sb.sync()
                si = sb.info()
                times[mode] = (si.sum_score_kernel_ms / max(si.timed_runs, 1), sb.fetch(), si.postings)
                sb.close()
            assert all(a.tobytes() == b_.tobytes() for a, b_ in zip(times[False][1], times[True][1])), "pruned single-term results differ"
            subset = {"queries": len(st_queries), "postings_per_query": times[True][2] / len(st_queries),
                      "kernel_ms_exhaustive": times[False][0], "kernel_ms_pruned": times[True][0]}
            eng.use_pruning(True)
        pruned_leg = {
            "what": "same batch, kernel only; single-term queries (found = the list's posting count) skip the 256-posting blocks whose maximum cannot enter their top-K; all other queries exhaustive",
            "kernel_ms": k3, "queries_per_s_kernel": Qr / (k3 * 1e-3) if k3 > 0 else 0.0,
            "single_term_queries": single, "single_term_queries_alone": subset, "build_s": build_s, "identical_results": same,
            "note": "not a roofline number: the bytes of skipped blocks are never read",
        }
        b3.close()
        eng.use_pruning(False)

    traffic_db = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic_db = json.load(f)
        except Exception:   # noqa: BLE001
            traffic_db = {}

    hbm_leg = None
    if n_gpus == 1 and args.config == "cfg5" and not args.no_hbm_leg and not args.kernel_only:
        hbm_leg = run_hbm_leg(args, nsbind, np, gen, seed, docs, K, flags, local_rank, traffic_db)

    if rank == 0:
        achieved = kinfo.algo_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        traffic = traffic_db.get(f"{args.config}_v{args.variant}_q{Qr}")
        value = (global_batch * args.steps / elapsed) if elapsed else (Qr * n_gpus * args.steps / kernel_elapsed)
        line = {
            "metric": "queries/sec at k=10 over 1M-doc index; achieved HBM GB/s vs peak" if args.config in ("cfg5", "cfg4")
                      else f"queries/sec ({args.config})",
            "value": value,
            "unit": "queries/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": (elapsed if elapsed else kernel_elapsed) / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": WORKLOAD_TEXT[args.config],
                "index": f"{nseg} segment(s) x {docs} docs, 65536-term Zipf vocabulary, {kinfo.postings / max(Qr, 1):.0f} postings/query",
                "global_batch": global_batch,
                "queries_per_gpu": Qr,
                "k": K,
                "parallelism": (f"query-sharded x{n_gpus}: index replicated, " + ("one global batch in contiguous shards" if strong else "a full batch per rank")
                                + (", ONE RCCL all-gather of the packed result blocks per step" if n_gpus > 1 else "")),
                "batches": "the steps rotate over 4 distinct batches of the law (batch 0 = the roofline leg's; seeds + 104729 i)",
                "timed_region": ("per step: ns_batch_prepare from host-resident term refs (regroup, work items, H2D) -> scoring + merge kernels -> "
                                 + ("results in host memory" if n_gpus == 1 else "all-gather of all ranks' results on the device")
                                 + f"; {args.depth} steps in flight per ctx" + ("" if args.no_overlap else ", alternating between two streams")) if elapsed else "kernel leg only (--kernel-only)",
                "kernel_variant": args.variant,
                "work_items": kinfo.n_items,
                "index_side_data": "skip tables for lists of >= n_docs/512 postings (4 B per list and 1024-doc cell, built at reload from the uploaded postings); no impact / packed streams in this leg",
                "term_scores": (f"shared inside the batch: the {kinfo.shared_lists} distinct lists ({kinfo.shared_postings} postings) the batch names are scored once per step by k_share_scores, "
                                f"in front of the scoring kernel and inside the timed region ({kinfo.postings / max(kinfo.shared_postings, 1):.0f} uses per distinct posting); nothing is kept between steps")
                               if shared else "in place: every posting scored once per query that names its list",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_share_scores + k_uscore" if (args.variant == 0 and shared) else "k_uscore" if args.variant == 0 else "k_dscore" if args.variant in (12, 13, 14, 15, 16, 17) else "k_tscore" if args.variant in (18, 19, 20) else "k_score",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "L2-miss bytes per launch from the builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --kernel-only` (x2 gfx950 FETCH_SIZE correction), profiles/r03/final_cfg5_pmc_traffic.csv; not measured in this run",
                "limited_by": ("instruction issue: this index (83 MB on the device) is cache resident — with the XCD-aware launch order and the shared term scores a step misses L2 for 0.15x its algorithmic bytes — "
                               "and the SIMDs' vector and scalar issue slots are 76 % / 76 % busy (profiles/r03/final_cfg5_pmc_sq.csv: SQ_ACTIVE_INST_VALU / SQ_INSTS_SALU vs SQ_BUSY_CYCLES; 83 % / 73 % when every posting is scored in place); see hbm_resident for the HBM-resident leg")
                              if args.config in ("cfg5", "cfg3", "cfg4") else "launch latency (a few tens of microseconds of work)",
                "algo_bytes_per_launch": int(kinfo.algo_bytes),
                "kernel_ms": score_ms,
                "all_kernels_ms": total_ms,
                "measured": "HIP events on the ctx stream around every scoring launch of the kernel leg (descriptors resident, one batch at a time)"
                            + ("; the span holds BOTH kernels of a step: k_share_scores (the batch's distinct lists scored once) and k_uscore" if shared else ""),
            },
            "kernel_only": {"value": Qr * n_gpus * args.steps / kernel_elapsed, "unit": "queries/s", "ms_per_step": kernel_elapsed / args.steps * 1e3,
                            "what": "device-only: the prepared batch re-run with descriptors resident in HBM (round 1's `value`)"},
            "host": host_info(),
            "library": lib_info(),
        }
        if n_gpus == 1 and args.cpu_seconds > 0 and flags == 0 and not args.kernel_only:
            cb, allcb = cpu_baselines(index_dir, queries, K, args.cpu_seconds)
            if cb is not None:
                line["cpu_baseline"] = cb
                line["cpu_baselines"] = allcb
        if hbm_leg is not None:
            line["hbm_resident"] = hbm_leg
        if in_place_leg is not None:
            line["in_place"] = in_place_leg
        if impact_leg is not None:
            line["impact_stream"] = impact_leg
        if pruned_leg is not None:
            line["pruned"] = pruned_leg
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
