#!/usr/bin/env python3
"""Micro-benchmark of the scoring kernel on synthetic query laws (diagnostic tool, GPU box only).

    python tools/law_bench.py [--variant V] [--split S] [--laws A,B,...]

Prints, per law: queries, postings/query, kernel ms, ns per posting-lane and algorithmic GB/s.
"""
import argparse
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import nsbind  # noqa: E402
import workloads  # noqa: E402

T = workloads.term_name


def laws():
    rng = random.Random(1)
    L = {}
    L["r1"] = ([T(1)] * 1024, 10)
    L["r2"] = ([T(2)] * 2048, 10)
    L["r4"] = ([T(4)] * 4096, 10)
    L["r8"] = ([T(8)] * 4096, 10)
    L["r16"] = ([T(16)] * 8192, 10)
    L["r32"] = ([T(32)] * 8192, 10)
    L["r100"] = ([T(100 + i % 50) for i in range(16384)], 10)
    L["r1000"] = ([T(1000 + i % 500) for i in range(16384)], 10)
    L["r1000x3"] = ([" ".join(T(rng.randint(1000, 3000)) for _ in range(3)) for _ in range(16384)], 10)
    L["r10000x3"] = ([" ".join(T(rng.randint(10000, 30000)) for _ in range(3)) for _ in range(16384)], 10)
    L["r1+r1000x2"] = ([T(1) + " " + " ".join(T(rng.randint(1000, 3000)) for _ in range(2)) for _ in range(1024)], 10)
    L["r8+r1000x2"] = ([T(8) + " " + " ".join(T(rng.randint(1000, 3000)) for _ in range(2)) for _ in range(4096)], 10)
    L["r1r2r3r4r5"] = ([" ".join(T(r) for r in (1, 2, 3, 4, 5))] * 512, 100)
    L["cfg5"] = (workloads.cfg5_queries(), 10)

    def rank_of(t):
        return workloads.WORDS.index(t) + 1 if t in workloads.WORDS else int(t[1:])
    q5 = workloads.cfg5_queries()
    # decomposition of cfg5: only the most frequent term of every query / queries without a hot term /
    # queries with exactly one hot term / with two or more
    L["cfg5_top1"] = ([min(q.split(), key=rank_of) for q in q5], 10)
    L["cfg5_nohot"] = ([q for q in q5 if min(rank_of(t) for t in q.split()) > 32], 10)
    L["cfg5_1hot"] = ([q for q in q5 if sum(1 for t in q.split() if rank_of(t) <= 32) == 1], 10)
    L["cfg5_1hot_multi"] = ([q for q in q5 if sum(1 for t in q.split() if rank_of(t) <= 32) == 1 and len(q.split()) > 1], 10)
    L["cfg5_2hot"] = ([q for q in q5 if sum(1 for t in q.split() if rank_of(t) <= 32) >= 2], 10)
    # cfg5's two-hot class split the way ns_batch_prepare classifies (df = 0.6 N / rank on the synthetic index)
    def cls_of(q):
        dfs = [600000.0 / rank_of(t) for t in q.split()]
        cost, cmax = sum(dfs), max(dfs)
        rest = cost - cmax
        if rest * 32 <= cmax:
            return "thin"
        if len(dfs) >= 2 and cost * 100 >= 1_000_000 * 25:
            return "tile"
        return "gen"
    for c in ("thin", "tile", "gen"):
        L["cfg5_2hot_" + c] = ([q for q in L["cfg5_2hot"][0] if cls_of(q) == c], 10)
        L["cfg5_" + c] = ([q for q in q5 if cls_of(q) == c], 10)
    L["cfg5_t2_gen"] = ([q for q in q5 if len(q.split()) == 2 and cls_of(q) == "gen"], 10)   # two-list groups of the general class: the merge body's
    L["r8r20"] = ([T(8) + " " + T(20)] * 4096, 10)
    L["r20r8"] = ([T(20) + " " + T(8)] * 4096, 10)
    L["r40r60"] = ([T(40 + i % 7) + " " + T(60 + i % 11) for i in range(8192)], 10)
    L["r8r300"] = ([T(8) + " " + T(300 + i % 50) for i in range(4096)], 10)
    L["cfg5_1hot_gen"] = ([q for q in L["cfg5_1hot"][0] if cls_of(q) == "gen"], 10)
    L["cfg5_nohot_gen"] = ([q for q in L["cfg5_nohot"][0] if cls_of(q) == "gen"], 10)
    L["cfg3"] = (workloads.cfg3_queries(), 100)
    L["cfg3_k10"] = (workloads.cfg3_queries(), 10)
    L["cfg5_seed7"] = (workloads.cfg5_queries(16384, 7), 10)
    for nq in (1, 8, 64, 256):
        L[f"cfg5_q{nq}"] = (workloads.cfg5_queries(256, 2005)[:nq], 10)
    L["hot5_q1"] = ([" ".join(T(r) for r in (1, 2, 3, 4, 5))], 10)
    L["cfg5_seed99_q32768"] = (workloads.cfg5_queries(32768, 99), 10)
    L["cfg5_q4096"] = (workloads.cfg5_queries(4096, 2005), 10)
    L["cfg5_q1024"] = (workloads.cfg5_queries(1024, 2005), 10)
    L["cfg5_q2048"] = (workloads.cfg5_queries(2048, 2005), 10)
    L["cfg5_q512"] = (workloads.cfg5_queries(512, 2005), 10)
    L["cfg3_k64"] = (workloads.cfg3_queries(), 64)
    L["cfg3_seed9"] = (workloads.cfg3_queries(4096, 9), 100)
    L["hot5_k10"] = ([" ".join(T(r) for r in (1, 2, 3, 4, 5))] * 1024, 10)
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--laws", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--impacts", action="store_true", help="build the optional impact streams first")
    ap.add_argument("--packed", type=int, default=0, help="build the packed posting streams first and read them in this mode (1: norms from the fp32 stream, 2: through the 16-bit norm index)")
    ap.add_argument("--prune", action="store_true", help="build block maxima and let single-term queries skip blocks (ns_ctx_use_pruning)")
    ap.add_argument("--no-skips", action="store_true", help="ignore the skip tables reload() built")
    ap.add_argument("--share", type=int, default=-1, help="ns_ctx_share_scores mode (0 never, 1 default rule, 2 always); -1: leave the library's default")
    ap.add_argument("--segments", type=int, default=1, help="segments of --docs docs each (20 x 1M docs = 1.1 GB of postings: beyond the 256 MiB Infinity Cache)")
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--qscale", type=float, default=1.0, help="keep only this fraction of every law's queries (every query scans every segment)")
    args = ap.parse_args()
    tmp = tempfile.TemporaryDirectory(prefix="ns_law_")
    idx = os.path.join(tmp.name, "index")
    nsbind.gen_index(idx, args.segments, args.docs, 65536, 1337, False)
    eng = nsbind.Engine(idx, 0)
    eng.set_tuning(args.variant, 0, args.split)
    if args.impacts:
        eng.build_impacts()
    if args.packed:
        eng.build_packed()
        eng.use_packed(args.packed)
    if args.no_skips:
        eng.use_skips(False)
    if args.share >= 0:
        eng.share_scores(args.share)
    if args.prune:
        eng.build_blockmax()
        eng.use_pruning(True)
    L = laws()
    # every list of ranks 1..4096 scanned by exactly one query: each posting byte is read once per launch
    if args.qscale != 1.0:
        L = {n: (qs[:max(1, int(len(qs) * args.qscale))], k) for n, (qs, k) in L.items()}
    L["scan_once"] = ([T(r) for r in range(1, 4097)], 10)
    names = [n for n in args.laws.split(",") if n] or list(L.keys())
    print(f"variant={args.variant} split={args.split} share={args.share} prune={args.prune} impacts={args.impacts} packed={args.packed} segments={args.segments} docs={args.docs} qscale={args.qscale}")
    print(f"{'law':>14} {'Q':>6} {'post/q':>9} {'items':>7} {'kern_ms':>9} {'ns/post':>8} {'GB/s':>8} {'frac':>6} {'all_ms':>8}")
    for n in names:
        qs, k = L[n]
        if not qs:
            print(f"{n:>14}      0")
            continue
        b = eng.prepare(qs, k)
        b.run(False)
        b.sync()
        for _ in range(args.reps):
            b.run(True)
        b.sync()
        inf = b.info()
        ms = inf.sum_score_kernel_ms / inf.timed_runs
        gbs = inf.algo_bytes / (ms * 1e-3) / 1e9
        print(f"{n:>14} {len(qs):>6} {inf.postings / len(qs):>9.0f} {inf.n_items:>7} {ms:>9.3f} {ms * 1e6 / inf.postings:>8.4f} {gbs:>8.0f} {gbs / 8000:>6.3f} {inf.sum_total_ms / inf.timed_runs:>8.3f}"
              + (f"   shared: {inf.shared_lists} lists, {inf.shared_postings} postings" if inf.shared_lists else ""))
        b.close()
    eng.close()


if __name__ == "__main__":
    main()
