#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REAL reference engine (oracle/_ref/ref_driver, built
from /root/reference in place by `make -C oracle ref`) on indexes produced by this repo's
deterministic generator.  Runs only in the build container (the reference never travels); the
fixtures it writes are data: generator parameters, SHA-256 of every generated index file, the query
texts and the reference's outputs (found, and per hit: segment index, docId, fp32 score bits).

    python tools/gen_golden.py
"""
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nsbind  # noqa: E402
import orc  # noqa: E402
import workloads  # noqa: E402

FIXTURES = {
    "small2": dict(n_segments=2, docs_per_segment=3000, vocab=2048, seed=1337, legacy=False),
    "legacy1": dict(n_segments=1, docs_per_segment=2000, vocab=512, seed=7, legacy=True),
    "mid1": dict(n_segments=1, docs_per_segment=20000, vocab=8192, seed=1337, legacy=False),
    "multi8": dict(n_segments=8, docs_per_segment=1500, vocab=1024, seed=99, legacy=False),
}

EDGE_QUERIES = [
    "covid",                       # BASELINE config 1: single hot term (large tie groups)
    "COVID",                       # case folding
    "covid covid",                 # duplicate terms are scored twice (src/api_engine.cpp:391-397)
    "virus vaccine",
    "the covid of a virus",        # stop-words and 1-char tokens dropped
    "covid-19: virus/vaccine?",    # punctuation splits tokens; "19" is not in the lexicon
    "zzzzunknown",                 # no term in any lexicon -> found 0, results []
    "the a of",                    # only stop-words -> early return without "found"
    "",                            # empty query -> early return
    "x y z",                       # only 1-char tokens -> early return
    "t000009 t000010 t000011 t000012 t000013 t000014 t000015 t000016 t000017",   # 9 terms
    "pandemic respiratory coronavirus patients infection vaccine virus covid",
    "t000100",
    "t000500 t000501",
    "covid\tvirus\x01vaccine",     # control bytes split tokens
    "caf\xc3\xa9 covid",           # bytes >= 0x80 split tokens
]


def sha256_tree(root):
    out = {}
    for d, _, files in sorted(os.walk(root)):
        for fn in sorted(files):
            p = os.path.join(d, fn)
            with open(p, "rb") as f:
                out[os.path.relpath(p, root)] = hashlib.sha256(f.read()).hexdigest()
    return out


def generated_queries(vocab, seed):
    rng = random.Random(seed)
    qs = []
    z = workloads._Zipf(1, min(5000, vocab))
    for _ in range(12):   # cfg3-like: 5 distinct zipf terms
        ranks = []
        while len(ranks) < 5:
            r = z.draw(rng)
            if r not in ranks:
                ranks.append(r)
        qs.append(" ".join(workloads.term_name(r) for r in ranks))
    qs += workloads.cfg5_queries(12, seed + 1, vocab)
    for _ in range(6):    # 2-term mid-frequency
        a, b = rng.randint(10, min(1000, vocab)), rng.randint(10, min(1000, vocab))
        qs.append(f"{workloads.term_name(a)} {workloads.term_name(b)}")
    return qs


def main():
    if not os.path.exists(orc.REF_DRIVER):
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle ref` where /root/reference is mounted")
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, p in FIXTURES.items():
        tmp = tempfile.mkdtemp(prefix="ns_golden_")
        try:
            idx = os.path.join(tmp, "index")
            total = nsbind.gen_index(idx, p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
            queries = [q for q in EDGE_QUERIES] + generated_queries(p["vocab"], 4242)
            # the driver's text protocol is line based: keep queries single-line
            queries = [q.replace("\n", " ") for q in queries]
            cases = []
            for k in (1, 10, 100, 0, 250):   # 0 and 250 exercise the clamp (src/api_engine.cpp:377)
                res = orc.run_ref_driver(idx, queries, k, tmp)
                cases.append({"k": k, "results": [{"found": r["found"], "hits": r["hits"]} for r in res]})
            fixture = {
                "name": name,
                "params": p,
                "total_postings": total,
                "sha256": sha256_tree(idx),
                "queries": queries,
                "cases": cases,
                "source": "cord19::Engine::search of /root/reference (g++ -O2), via oracle/_ref/ref_driver",
            }
            with open(os.path.join(outdir, f"{name}.json"), "w") as f:
                json.dump(fixture, f, separators=(",", ":"))
            print(name, "postings", total, "queries", len(queries), "bytes", os.path.getsize(os.path.join(outdir, f"{name}.json")))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
