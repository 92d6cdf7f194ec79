#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of k_wscore (needs the -DNS_STAMP build copied over
nextsearch-api_amd/libnextsearch_hip.so on the GPU box)."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import nsbind, law_bench
L = nsbind.hip_lib()
L.ns_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
eng = nsbind.Engine(idx, 0)
eng.set_tuning(int(sys.argv[1]) if len(sys.argv) > 1 else 0, 0, 0)
laws = law_bench.laws()
names = ["0 setup", "1 term prologue", "2 round size + load issue", "3 (prefetch) + load wait", "4 masks+bm25+LDS reads", "5 LDS writes",
         "6 round end / next-doc", "7 term epilogue", "8 next header", "9 read-back scan", "10 shrink/loop", "11 final"]
for n in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["cfg5_tile"]):
    qs, k = laws[n]
    b = eng.prepare(qs, k)
    b.run(False); b.sync()
    out = (C.c_uint64 * 32)()
    L.ns_debug_stamps(out, 1)
    b.run(True); b.sync()
    L.ns_debug_stamps(out, 1)
    inf = b.info()
    tot = sum(out[i] for i in range(12)); waves = out[31]
    nb = inf.postings / 64.0
    print(f"{n}: kernel {inf.last_score_kernel_ms:.3f} ms, tile-body waves {waves}, cycles per 64 postings {tot/nb:.0f}")
    for i in range(12):
        print(f"    {names[i]:>30}: {100.0*out[i]/tot:5.1f}%  ({out[i]/nb:8.0f} cyc/chunk)")
    b.close()
