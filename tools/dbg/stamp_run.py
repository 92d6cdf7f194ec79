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
names = ["mark7:other", "p0:wait loads+hi", "p1:take/consumed", "p2:issue next", "p3:bm25+adds", "p4:readout scan", "p5:collect", "p6:reset"]
for n in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["r1", "r8", "r32", "r1000x3", "cfg5"]):
    qs, k = laws[n]
    b = eng.prepare(qs, k)
    b.run(False); b.sync()
    out = (C.c_uint64 * 16)()
    L.ns_debug_stamps(out, 1)
    b.run(True); b.sync()
    L.ns_debug_stamps(out, 1)
    inf = b.info()
    tot = sum(out[i] for i in range(8)); waves = out[15]
    nb = inf.postings / 256.0
    print(f"{n}: kernel {inf.last_score_kernel_ms:.3f} ms, waves {waves}, cycles/wave {tot/max(waves,1):.0f}, est cycles/batch {tot/nb:.0f}")
    order = [1, 2, 3, 4, 5, 6, 7, 0]
    lab = dict(zip([7,0,1,2,3,4,5,6], names))
    for i in [0,1,2,3,4,5,6,7]:
        print(f"    {lab[i]:>22}: {100.0*out[i]/tot:5.1f}%  ({out[i]/nb:8.0f} cyc/batch)")
    b.close()
