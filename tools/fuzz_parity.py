#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle (test infrastructure; GPU box only):
random index shapes, query laws, K, OR/AND, work-splitting knobs, with and without impact streams, packed streams and skip tables.
Stops at the first mismatch and prints the case; prints a summary line otherwise."""
import argparse
import os
import random
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nsbind  # noqa: E402
import orc  # noqa: E402
import workloads  # noqa: E402


def same(gpu, ora):
    gh, gn, gf, gu = gpu
    oh, on, of, ou = ora
    if not np.array_equal(gu.astype(bool), ou.astype(bool)):
        return "usable"
    for q in range(len(on)):
        if not ou[q]:
            continue
        if int(gf[q]) != int(of[q]):
            return f"found q{q}: {gf[q]} vs {of[q]}"
        if int(gn[q]) != int(on[q]):
            return f"nhits q{q}"
        n = int(on[q])
        g, o = gh[q, :n], oh[q, :n]
        if not (np.array_equal(g["doc"], o["doc"]) and np.array_equal(g["seg"], o["seg"]) and np.array_equal(g["score"].view(np.uint32), o["score"].view(np.uint32))):
            return f"hits q{q}"
    return None


def run(seconds, seed, max_cases=None, verbose=True):
    """Returns (batches checked, None) or (batches checked, description of the first mismatch)."""
    import shutil
    rng = random.Random(seed)
    t0 = time.time()
    last = t0
    cases = 0
    tmp = tempfile.mkdtemp(prefix="ns_fuzz_")
    try:
        return _run(rng, t0, last, cases, tmp, seconds, max_cases, verbose)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _run(rng, t0, last, cases, tmp, seconds, max_cases, verbose):
    while time.time() - t0 < seconds and (max_cases is None or cases < max_cases):
        nseg = rng.choice([1, 1, 2, 3, 5])
        docs = rng.choice([300, 2000, 9000, 40000, 120000])
        vocab = rng.choice([64, 512, 4096, 16384])
        seed = rng.randint(1, 10**6)
        idx = os.path.join(tmp, f"i{cases}")
        nsbind.gen_index(idx, nseg, docs, vocab, seed, rng.random() < 0.15 and nseg == 1)
        eng, ora = nsbind.Engine(idx, 0), orc.Oracle(idx)
        try:
            for rep in range(4):
                nq = rng.choice([1, 3, 17, 64, 300])
                law = rng.choice(["cfg5", "cfg3", "pairs", "dups", "many"])
                qs = []
                for _ in range(nq):
                    if law == "cfg5":
                        qs.append(workloads.cfg5_queries(1, rng.randint(1, 10**6), vocab)[0])
                    elif law == "cfg3":
                        qs.append(" ".join(workloads.term_name(rng.randint(1, min(vocab, 40))) for _ in range(5)))
                    elif law == "pairs":
                        qs.append(workloads.term_name(rng.randint(1, min(vocab, 12))) + " " + workloads.term_name(rng.randint(1, vocab)))
                    elif law == "dups":
                        t = workloads.term_name(rng.randint(1, min(vocab, 30)))
                        qs.append(" ".join([t] * rng.randint(2, 4) + [workloads.term_name(rng.randint(1, vocab))]))
                    else:
                        qs.append(" ".join(workloads.term_name(rng.randint(1, vocab)) for _ in range(rng.randint(9, 40))))
                k = rng.choice([1, 3, 10, 33, 64, 100])
                flags = rng.choice([0, 0, 0, nsbind.NS_FLAG_AND])
                tune = rng.choice([(0, 0, 0), (0, 0, 0), (0, 4096, 0), (0, 1, 1 << 30), (0, 20000, 700), (0, 1, 300)])
                eng.set_tuning(*tune)
                imp = rng.random() < 0.4
                if imp:
                    eng.build_impacts()
                eng.use_impacts(imp)
                pk = rng.choice([0, 0, 1, 2])
                if pk:
                    eng.build_packed()
                eng.use_packed(pk)
                skips = rng.random() < 0.75     # reload() builds skip tables for the frequent lists; searches may ignore them
                eng.use_skips(skips)
                merge = rng.random() < 0.7      # two-list groups: merge body or driver-stream body
                eng.use_merge(merge)
                prune = rng.random() < 0.3      # single-term queries: block-max pruning
                if prune:
                    eng.build_blockmax()
                eng.use_pruning(prune)
                share = rng.choice([0, 2, 2])   # term scores computed once per distinct list of the batch (until the engine has an optional impact stream)
                eng.share_scores(share)
                bad = same(eng.search_batch(qs, k, flags), ora.search_batch(qs, k, flags, threads=8))
                if bad:
                    return cases, (f"MISMATCH {bad}: index(nseg={nseg}, docs={docs}, vocab={vocab}, seed={seed}) law={law} nq={nq} k={k} "
                                   f"flags={flags} tune={tune} impacts={imp} packed={pk} skips={skips} merge={merge} prune={prune} share={share} queries={qs[:5]}")
                cases += 1
                if verbose and time.time() - last > 30:
                    last = time.time()
                    print(f"... {cases} batches so far ({last - t0:.0f} s)", flush=True)
        finally:
            eng.close()
            ora.close()
            import shutil
            shutil.rmtree(idx, ignore_errors=True)
    return cases, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    t0 = time.time()
    cases, bad = run(args.seconds, args.seed)
    if bad:
        print(bad)
        sys.exit(1)
    print(f"fuzz: {cases} batches equal to the oracle in {time.time() - t0:.0f} s (seed {args.seed})")


if __name__ == "__main__":
    main()
