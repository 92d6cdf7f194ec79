#!/usr/bin/env python3
"""Diagnostic: event counts of the driver-stream body (needs the -DNS_STAMP build copied over
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
laws = law_bench.laws()
names = ["items", "super-batches", "sb with foreign", "foreign chunks", "claim iterations", "rmw term iters", "driver rounds",
         "lookup chunks", "lookup slow entries", "hit branches", "-", "-", "foreign postings", "driver postings", "foreign window postings", "-"]
for n in sys.argv[1].split(","):
    qs, k = laws[n]
    b = eng.prepare(qs, k)
    out = (C.c_uint64 * 32)()
    L.ns_debug_stamps(out, 1)
    b.run(True); b.sync()
    L.ns_debug_stamps(out, 1)
    inf = b.info()
    print(f"{n}: postings {inf.postings}, dscore items {out[31]}")
    for i in range(16):
        if names[i] != "-":
            print(f"    {names[i]:>24}: {out[i]:>12}  ({out[i] / max(out[1], 1):8.2f} per super-batch)")
    b.close()
