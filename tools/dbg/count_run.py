#!/usr/bin/env python3
"""Diagnostic: event counts of the driver-stream body per query law (GPU box only).  Needs the counting build:
    make -C nextsearch-api_amd count        (here; the .so travels with the snapshot)
    on the GPU box (a scratch copy of the tree): cp nextsearch-api_amd/libnextsearch_hip_count.so nextsearch-api_amd/libnextsearch_hip.so
    python3 tools/dbg/count_run.py cfg5_gen,cfg5_thin,cfg5
"""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import nsbind, law_bench
L = nsbind.hip_lib()
L.ns_debug_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int]
L.ns_debug_tile_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int]
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
eng = nsbind.Engine(idx, 0)
laws = law_bench.laws()
names = ["items (driver-stream body)", "super-batches", "super-batches with foreign postings", "foreign postings LOADED (windows)", "foreign postings consumed",
         "foreign chunks", "claim iterations", "rmw term passes", "driver rounds (256 loaded each)", "driver postings consumed", "terms (sum over items)",
         "active foreign terms (sum over sb)", "driver chunks with postings", "driver chunks that probed the table", "driver postings that hit the table", "foreign entries placed WITHOUT a claim (pass A)"]
for n in (sys.argv[1] if len(sys.argv) > 1 else "cfg5_gen").split(","):
    qs, k = laws[n]
    b = eng.prepare(qs, k)
    out = (C.c_uint64 * 20)()
    tout = (C.c_uint64 * 12)()
    L.ns_debug_counters(out, 1); L.ns_debug_tile_counters(tout, 1)
    b.run(True); b.sync()
    L.ns_debug_counters(out, 1); L.ns_debug_tile_counters(tout, 1)
    inf = b.info()
    sb = max(out[1], 1)
    print(f"{n}: postings {inf.postings}, kernel {inf.last_score_kernel_ms:.3f} ms")
    for i in range(16):
        print(f"    {names[i]:>40}: {out[i]:>12}  ({out[i] / sb:8.2f} per super-batch)")
    print(f"    foreign window utilisation {out[4] / max(out[3], 1):.3f}; lanes used in foreign chunks {out[4] / max(out[5] * 64, 1):.3f}; "
          f"driver round utilisation {out[9] / max(out[8] * 256, 1):.3f}; lanes used in driver chunks {out[9] / max(out[12] * 64, 1):.3f}; "
          f"foreign share of consumed postings {out[4] / max(out[4] + out[9], 1):.3f}")
    if out[0] and out[17]:
        print(f"    driver-stream body, shader clocks per item: whole {out[17] / out[0]:.0f}, set-up (term table, range searches, first window plan) {out[16] / out[0]:.0f}, final shrink + rows out {out[18] / out[0]:.0f}")
    if tout[0]:
        t = max(tout[1], 1)
        print(f"    doc-tile body: items {tout[0]}, tiles {tout[1]}, (term, tile) visits {tout[2]} ({tout[2] / t:.2f} per tile), rounds {tout[3]} ({tout[3] / max(tout[2], 1):.2f} per visit), "
              f"chunks loaded {tout[4]} ({tout[4] / t:.2f} per tile), postings taken {tout[5]} ({tout[5] / t:.1f} per tile; lanes used {tout[5] / max(tout[4] * 64, 1):.3f}), terms per item {tout[6] / tout[0]:.2f}")
        if tout[9]:
            print(f"    doc-tile body, shader clocks: item {tout[7] / tout[0]:.0f} per item ({tout[7] / max(tout[5], 1) * 64:.0f} per 64 postings); full rounds {tout[9]}: "
                  f"issue -> data {tout[8] / tout[9]:.0f}, whole round {tout[11] / tout[9]:.0f}; tile read-back (+ shrink) {tout[10] / t:.0f} per tile")
    b.close()
