#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REAL reference engine (oracle/_ref/ref_driver, built
from /root/reference in place by `make -C oracle ref`) on indexes produced by this repo's
deterministic generator.  Runs only in the build container (the reference never travels); the
fixtures it writes are data: generator parameters, SHA-256 of every generated index file, the query
texts and the reference's outputs (found, and per hit: segment index, docId, fp32 score bits).

    python tools/gen_golden.py
"""
import base64
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile
import time

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


META_FIXTURE = dict(n_segments=2, docs_per_segment=3000, vocab=2048, seed=1337, legacy=False, meta_seed=11)
META_QUERIES = ["covid virus", "vaccine", "t000100 t000101 covid", "pandemic respiratory", "t000500", "the of", "",
                "zzzzunknown", "t000300 t000020", "coronavirus patients infection", "t001000 t001001 t001002", "COVID-19 vaccine"]


def run_ref_json(index_dir, queries, k, workdir):
    """ref_driver json: the reference's own Engine::search(...).dump(2) per query."""
    import subprocess
    qpath = os.path.join(workdir, "queries_json.txt")
    opath = os.path.join(workdir, "ref_json.txt")
    with open(qpath, "w") as f:
        f.write("\n".join(queries) + "\n")
    subprocess.check_call([orc.REF_DRIVER, "json", index_dir, qpath, str(k), opath], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    data = open(opath, "rb").read()
    out, pos = [], 0
    while pos < len(data):
        nl = data.index(b"\n", pos)
        assert data[pos:pos + 2] == b"J "
        n = int(data[pos + 2:nl])
        out.append(data[nl + 1:nl + 1 + n].decode("utf-8"))
        pos = nl + 1 + n + 1
    return out[:len(queries)]


def make_meta_fixture(outdir):
    p = META_FIXTURE
    tmp = tempfile.mkdtemp(prefix="ns_golden_")
    try:
        idx = os.path.join(tmp, "index")
        nsbind.gen_index(idx, p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
        csv = workloads.metadata_csv(p["n_segments"] * p["docs_per_segment"], p["meta_seed"])
        with open(os.path.join(idx, "metadata.csv"), "wb") as f:
            f.write(csv)
        cases = []
        for k in (5, 1):
            cases.append({"k": k, "json": run_ref_json(idx, META_QUERIES + [""], k, tmp)[:len(META_QUERIES)]})
        fixture = {"name": "meta1", "params": p, "metadata_csv_sha256": hashlib.sha256(csv).hexdigest(), "queries": META_QUERIES,
                   "cases": cases,
                   "source": "cord19::Engine::search(...).dump(2) of /root/reference (g++ -O2, nlohmann/json 3.1.1), via oracle/_ref/ref_driver json"}
        with open(os.path.join(outdir, "meta1.json"), "w") as f:
            json.dump(fixture, f, separators=(",", ":"))
        print("meta1 bytes", os.path.getsize(os.path.join(outdir, "meta1.json")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_invert_fixture(outdir):
    """Index inversion: a small forward.bin / terms.bin pair (committed as base64: a fixture is
    inputs + expected outputs) and the sha256 of every file the REAL `lexicon` tool (oracle/_ref/lexicon)
    writes for it."""
    import base64
    import hashlib
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import forward_gen
    import invert_oracle
    tool = os.path.join(ROOT, "oracle", "_ref", "lexicon")
    if not os.path.exists(tool):
        sys.exit("oracle/_ref/lexicon missing: run `make -C oracle ref` where /root/reference is mounted")
    tmp = tempfile.mkdtemp(prefix="ns_golden_inv_")
    try:
        seg = os.path.join(tmp, "seg")
        pairs = forward_gen.write_inputs(seg, 400, 900, 18, 20261)
        inputs = {f: base64.b64encode(open(os.path.join(seg, f), "rb").read()).decode() for f in ("terms.bin", "forward.bin")}
        subprocess.run([tool, seg], check=True, stderr=subprocess.DEVNULL)
        outs = {}
        for f in invert_oracle.output_files():
            b = open(os.path.join(seg, f), "rb").read()
            outs[f] = {"bytes": len(b), "sha256": hashlib.sha256(b).hexdigest()}
        with open(os.path.join(outdir, "invert1.json"), "w") as fh:
            json.dump({"what": "oracle/_ref/lexicon (the reference's src/lexicon.cpp, built as it lies) on the inputs below",
                       "generator": "forward_gen.write_inputs(seg, 400, 900, 18, 20261)", "pairs": pairs,
                       "inputs_base64": inputs, "outputs": outs}, fh, indent=0)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SEM_FIXTURE = dict(n_segments=2, docs_per_segment=2500, vocab=1536, seed=23, legacy=False, emb_dim=24, emb_seed=5)


def sem_queries(vocab):
    rng = random.Random(99)
    qs = ["covid vaccine", "covid covid", "virus", "the of a", "", "zzzz nothing here", "COVID-19 Vaccine!"]
    for _ in range(25):
        n = rng.choice([1, 2, 2, 3, 4])
        qs.append(" ".join(workloads.term_name(rng.randint(1, vocab)) for _ in range(n)))
    qs.append(workloads.term_name(40) + " " + workloads.term_name(40) + " zz_notindexed_5")
    return qs


def make_sem_fixture(outdir):
    """Semantic expansion: what the REAL reference does with an embeddings file next to the index —
    the loaded table's shape, the weighted terms Engine::search scores (order and weight bits), and the hits."""
    import subprocess
    p = SEM_FIXTURE
    tmp = tempfile.mkdtemp(prefix="ns_golden_sem_")
    try:
        idx = os.path.join(tmp, "index")
        nsbind.gen_index(idx, p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
        emb = workloads.embeddings_text(p["vocab"], p["emb_dim"], p["emb_seed"])
        with open(os.path.join(idx, "embeddings.vec"), "wb") as f:
            f.write(emb)
        queries = sem_queries(p["vocab"])
        qpath, opath = os.path.join(tmp, "q.txt"), os.path.join(tmp, "e.txt")
        with open(qpath, "w") as f:
            f.write("\n".join(queries) + "\n")
        subprocess.check_call([orc.REF_DRIVER, "expand", idx, qpath, "10", opath], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(opath).read().split("\n")
        _, enabled, rows, dim = lines[0].split()
        pos, expands = 1, []
        for _ in queries:
            n = int(lines[pos].split()[1])
            expands.append([[ln.split("\t")[0], int(ln.split("\t")[1], 16)] for ln in lines[pos + 1: pos + 1 + n]])
            pos += 1 + n
        cases = []
        for k in (10, 100):
            cases.append({"k": k, "results": orc.run_ref_driver(idx, queries, k, tmp)})
        with open(os.path.join(outdir, "sem1.json"), "w") as f:
            json.dump({"name": "sem1", "params": p, "embeddings_sha256": hashlib.sha256(emb).hexdigest(), "table": {"enabled": int(enabled), "rows": int(rows), "dim": int(dim)},
                       "queries": queries, "expand": expands, "cases": cases,
                       "source": "cord19::Engine (reload + sem.expand + search) of /root/reference with embeddings.vec in the index directory, via oracle/_ref/ref_driver expand|search"},
                      f, separators=(",", ":"))
        print("sem1 bytes", os.path.getsize(os.path.join(outdir, "sem1.json")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SEMLOAD_PARAMS = dict(n_segments=1, docs_per_segment=300, vocab=48, seed=7, legacy=False)


def semload_text():
    """A small embeddings file full of the spellings that tell one number reader from another (the reference reads the
    values with operator>>(float)): signs, bare points, exponents, numbers glued together, text in the middle of a
    line, a header look-alike that is not the first line, CR line ends, a repeated word, an all-zero vector."""
    T = workloads.term_name
    base = ["0.5", "-0.25", "1", "2.5e-1", "-3E+0", ".75", "4.", "+0.125", "1e1", "0.001", "7", "-8.5"]
    L = []
    L.append("  48   12  ")                                      # header (blanks around it)
    L.append(T(1) + " " + " ".join(base))
    L.append(T(2) + "\t" + "\t".join(reversed(base)) + "\r")    # tabs, CR at the end
    L.append("")
    L.append(T(3) + " 1 2 3 4 5 6 7 8 9 10 11 abc 12")           # stops at abc: 11 values -> another dimension, skipped
    L.append(T(4) + " 1 2 3 4 5 6 7 8 9 10 11 12 xyz 13")        # stops at xyz: 12 values, kept
    L.append(T(5) + " 1.5-3 2 3 4 5 6 7 8 9 10 11")              # "1.5-3" reads as 1.5 then -3: 12 values
    L.append(T(6) + " 1e 2 3 4 5 6 7 8 9 10 11 12")              # "1e" is not a number: no values at all
    L.append(T(7) + " 1e+ 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(8) + " . 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(9) + " 0 0 0 0 0 0 0 0 0 0 0 0")                  # zero vector: stays zero
    L.append(T(10) + " 1e39 2 3 4 5 6 7 8 9 10 11 12")           # overflows a float: extraction fails
    L.append(T(11) + " 1e-46 2 3 4 5 6 7 8 9 10 11 12")          # underflows to zero: fine
    L.append(T(12) + " 0x10 2 3 4 5 6 7 8 9 10 11 12")           # "0" then "x10" stops
    L.append(T(13) + " 1,5 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(14) + " inf 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(15) + " 00012.50 -0 +.5e1 1.e2 1.5E-2 3 4 5 6 7 8 9")
    L.append("12 12")                                            # looks like a header, but is not the first line
    L.append(T(1) + " 9 8 7 6 5 4 3 2 1 0 1 2")                  # repeated word: a new row, the name keeps its first row
    L.append("notinlexicon 1 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(16) + " 1 2 3 4 5 6 7 8 9")                       # < 10 values
    L.append(T(17) + " 1..2 2 3 4 5 6 7 8 9 10 11 12")           # "1." then ".2"
    L.append(T(18) + " 1e5e2 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(19) + " -+1 2 3 4 5 6 7 8 9 10 11 12")
    L.append(T(20) + " 3.4028235e38 -3.4028235e38 1.17549435e-38 16777217 0.1 0.2 0.3 0.7 1e-7 123456789 5 6")
    return ("\n".join(L) + "\n" + T(21) + " 12 11 10 9 8 7 6 5 4 3 2 1").encode()   # last line without a newline


def make_semload_fixture(outdir):
    """What the REAL reference's SemanticIndex::load_from_text makes of semload_text(): rows, names, fp32 bits."""
    import subprocess
    p = SEMLOAD_PARAMS
    tmp = tempfile.mkdtemp(prefix="ns_golden_semload_")
    try:
        idx = os.path.join(tmp, "index")
        nsbind.gen_index(idx, p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
        emb = semload_text()
        with open(os.path.join(idx, "embeddings.vec"), "wb") as f:
            f.write(emb)
        qpath, opath = os.path.join(tmp, "q.txt"), os.path.join(tmp, "t.txt")
        open(qpath, "w").write("covid\n")
        subprocess.check_call([orc.REF_DRIVER, "semtable", idx, qpath, "10", opath], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(opath).read().split("\n")
        _, enabled, rows, dim = lines[0].split()
        table = [[ln.split(" ")[0], [int(x, 16) for x in ln.split(" ")[1:]]] for ln in lines[1:1 + int(rows)]]
        with open(os.path.join(outdir, "semload1.json"), "w") as f:
            json.dump({"name": "semload1", "params": p, "embeddings_b64": base64.b64encode(emb).decode(), "enabled": int(enabled), "dim": int(dim), "table": table,
                       "source": "cord19::SemanticIndex::load_from_text of /root/reference (through Engine::reload), dumped by oracle/_ref/ref_driver semtable"},
                      f, separators=(",", ":"))
        print("semload1: rows", rows, "dim", dim, "bytes", os.path.getsize(os.path.join(outdir, "semload1.json")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_fullsize_fixture(outdir, procs=8):
    """BASELINE configs 2-5 at FULL size (1 M docs; 16384 / 4096 / 4096 / 1024 queries): digests of the whole batches'
    answers.  `exact`: the oracle (oracle/bm25_oracle.c, itself pinned by the goldens above).  `ties`: the REAL reference
    (oracle/_ref/ref_driver, `procs` processes on contiguous slices of the batch), for the OR configs — and the
    generator asserts that the oracle's own tie-invariant digests equal the reference's before it writes anything."""
    import subprocess
    import numpy as np
    fixture = {"name": "fullsize", "block": 1024, "configs": {},
               "source": "exact: oracle/bm25_oracle.c over the full batches; ties: cord19::Engine::search of /root/reference over the full batches (oracle/_ref/ref_driver)"}
    tmp = tempfile.mkdtemp(prefix="ns_golden_full_")
    try:
        made = {}
        for cfg in ("cfg2", "cfg3", "cfg4", "cfg5"):
            gen, Q, K, flags, (nseg, docs) = workloads.WORKLOADS[cfg]
            if (nseg, docs) not in made:
                idx = os.path.join(tmp, f"i_{nseg}_{docs}")
                nsbind.gen_index(idx, nseg, docs, 65536, 1337, False)
                made[(nseg, docs)] = idx
            idx = made[(nseg, docs)]
            queries = gen(Q)
            ora = orc.Oracle(idx)
            t0 = time.time()
            hits, nhits, found, usable = ora.search_batch(queries, K, flags, threads=procs)
            ora.close()
            assert usable.all()
            dig = orc.batch_digests(hits, nhits, found)
            entry = {"queries": Q, "k": K, "flags": flags, "index": [nseg, docs], "exact": dig["exact"], "oracle_seconds": round(time.time() - t0, 1)}
            if flags == 0:   # the reference has no conjunctive mode
                t0 = time.time()
                per = (Q + procs - 1) // procs
                jobs = []
                for i in range(procs):
                    sub = queries[i * per:(i + 1) * per]
                    if not sub:
                        continue
                    wd = os.path.join(tmp, f"w_{cfg}_{i}")
                    os.makedirs(wd)
                    qp, op = os.path.join(wd, "q.txt"), os.path.join(wd, "o.txt")
                    open(qp, "w").write("\n".join(sub) + "\n")
                    jobs.append((subprocess.Popen([orc.REF_DRIVER, "search", idx, qp, str(K), op], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), op, len(sub)))
                r_hits = np.zeros((Q, K), dtype=orc.HIT_DTYPE)
                r_nhits = np.zeros(Q, dtype=np.uint32)
                r_found = np.zeros(Q, dtype=np.uint64)
                q = 0
                for pr, op, n in jobs:
                    assert pr.wait() == 0
                    res = orc.parse_driver_output(op)
                    assert len(res) == n
                    for r in res:
                        r_found[q] = r["found"]
                        r_nhits[q] = len(r["hits"])
                        for j, (sg, dc, bits) in enumerate(r["hits"]):
                            r_hits[q, j] = (np.array([bits], dtype=np.uint32).view(np.float32)[0], sg, dc)
                        q += 1
                assert q == Q
                rdig = orc.batch_digests(r_hits, r_nhits, r_found)
                assert rdig["ties"] == dig["ties"], f"{cfg}: the oracle's answers differ from the real reference's over the full batch"
                entry["ties"] = rdig["ties"]
                entry["reference_seconds"] = round(time.time() - t0, 1)
            fixture["configs"][cfg] = entry
            print(cfg, {k: v for k, v in entry.items() if k not in ("exact", "ties")}, flush=True)
        with open(os.path.join(outdir, "fullsize.json"), "w") as f:
            json.dump(fixture, f, separators=(",", ":"))
        print("fullsize bytes", os.path.getsize(os.path.join(outdir, "fullsize.json")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def segwriter_spec(n_docs=150, seed=31):
    """Logical documents for the reference's SegmentWriter: cord_uid, title, json_relpath, doc_len, term:tf ..."""
    rng = random.Random(seed)
    vocab = ["covid", "virus", "vaccine", "protein", "cell"] + [workloads.term_name(r) for r in range(9, 400)]
    lines = []
    for d in range(n_docs):
        n = rng.choice([0, 1, 3, 8, 15, 30]) if d % 17 else 0          # some documents hold no terms
        terms = rng.sample(vocab, min(n, len(vocab)))
        tfs = ["%s:%d" % (t, 1 + int(rng.expovariate(0.7))) for t in terms]
        doc_len = sum(int(x.split(":")[1]) for x in tfs) + rng.randint(0, 40)
        lines.append("uid%05d\tTitle of %d\tdocs/%d.json\t%d\t%s" % (d, d, d, doc_len, " ".join(tfs)))
    return "\n".join(lines) + "\n"


def make_segwriter_fixture(outdir):
    """Tier T0: files written by the REFERENCE's own SegmentWriter (include/segment_writer.hpp) for a
    committed logical input — what this repo's loader must read and its inversion step must reproduce."""
    import base64
    import subprocess
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_segwriter")
    if not os.path.exists(tool):
        sys.exit("oracle/_ref/ref_segwriter missing: run `make -C oracle ref` where /root/reference is mounted")
    tmp = tempfile.mkdtemp(prefix="ns_golden_sw_")
    try:
        spec = segwriter_spec()
        sp = os.path.join(tmp, "spec.tsv")
        with open(sp, "w") as f:
            f.write(spec)
        idx = os.path.join(tmp, "index")
        os.makedirs(idx)
        subprocess.run([tool, sp, idx], check=True)
        seg = os.path.join(idx, "segments", "seg_000000")
        files = {}
        for name in sorted(os.listdir(seg)):
            b = open(os.path.join(seg, name), "rb").read()
            files[name] = {"bytes": len(b), "sha256": hashlib.sha256(b).hexdigest()}
        keep = {n: base64.b64encode(open(os.path.join(seg, n), "rb").read()).decode() for n in ("stats.bin", "docs.bin", "forward.bin", "terms.bin")}
        queries = ["covid", "virus vaccine", "protein cell covid", workloads.term_name(20) + " " + workloads.term_name(77), "nothinghere"]
        cases = [{"k": 10, "results": orc.run_ref_driver(idx, queries, 10, tmp)}]
        with open(os.path.join(outdir, "segwriter1.json"), "w") as f:
            json.dump({"what": "SegmentWriter::write_segment of /root/reference (include/segment_writer.hpp) on `spec`, via oracle/_ref/ref_segwriter; "
                               "`results`: the reference engine's search over that index (oracle/_ref/ref_driver)",
                       "spec": spec, "files": files, "inputs_base64": keep, "queries": queries, "cases": cases}, f, indent=0)
        print("segwriter1 bytes", os.path.getsize(os.path.join(outdir, "segwriter1.json")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_cache_fixture(outdir):
    """Search-result cache (src/api_engine.cpp:190-250): the reference's SECOND answer to each query (cache kept)."""
    import subprocess
    p = FIXTURES["small2"]
    tmp = tempfile.mkdtemp(prefix="ns_golden_cache_")
    try:
        idx = os.path.join(tmp, "index")
        nsbind.gen_index(idx, p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
        queries = ["covid", "covid vaccine", "the of", "zzzzunknown", "covid"]
        qpath, opath = os.path.join(tmp, "q.txt"), os.path.join(tmp, "o.txt")
        with open(qpath, "w") as f:
            f.write("\n".join(queries) + "\n")
        subprocess.check_call([orc.REF_DRIVER, "json2", idx, qpath, "5", opath], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(opath, "rb").read()
        out, pos = [], 0
        while data[pos:pos + 2] == b"J ":
            nl = data.index(b"\n", pos)
            n = int(data[pos + 2:nl])
            out.append(data[nl + 1:nl + 1 + n].decode("utf-8"))
            pos = nl + 1 + n + 1
        entries = int(data[pos + 2:data.index(b"\n", pos)])
        with open(os.path.join(outdir, "cache1.json"), "w") as f:
            json.dump({"name": "cache1", "params": p, "k": 5, "queries": queries, "second_answers": out, "cache_entries": entries,
                       "source": "cord19::Engine::search called twice per query, second dump(2), via oracle/_ref/ref_driver json2"}, f, separators=(",", ":"))
        print("cache1 bytes", os.path.getsize(os.path.join(outdir, "cache1.json")), "entries", entries)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "cache":
        make_cache_fixture(os.path.join(ROOT, "tests", "golden"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "segwriter":
        make_segwriter_fixture(os.path.join(ROOT, "tests", "golden"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        make_fullsize_fixture(os.path.join(ROOT, "tests", "golden"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "semload":
        make_semload_fixture(os.path.join(ROOT, "tests", "golden"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sem":
        make_sem_fixture(os.path.join(ROOT, "tests", "golden"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "invert":
        make_invert_fixture(os.path.join(ROOT, "tests", "golden"))
        return
    if not os.path.exists(orc.REF_DRIVER):
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle ref` where /root/reference is mounted")
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    make_meta_fixture(outdir)
    if len(sys.argv) > 1 and sys.argv[1] == "meta":
        return
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
