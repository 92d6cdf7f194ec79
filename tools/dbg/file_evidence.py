#!/usr/bin/env python3
"""Files the summaries of gpurun_out/<dir> (tools/collect_evidence.sh) under profiles/<round>/ and prints the numbers DESIGN.md quotes.
    python3 tools/dbg/file_evidence.py <dir> <round e.g. r02>"""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, "gpurun_out", sys.argv[1]); RND = sys.argv[2] if len(sys.argv) > 2 else "r03"
P = os.path.join(ROOT, "profiles", RND); os.makedirs(P, exist_ok=True)
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path))

def counters(path, name, key="uscore"):
    rows = list(csv.DictReader(open(path)))
    return [(r["Dispatch_Id"], r["Kernel_Name"], float(r["Counter_Value"])) for r in rows if key in r["Kernel_Name"] and r["Counter_Name"] == name]

for c in ["cfg5", "cfg2", "cfg3", "cfg4"]:
    f = f"{O}/bench_{c}.json"
    if os.path.exists(f) and os.path.getsize(f) > 10:
        shutil.copy(f, f"{P}/final_bench_{c}.json")
        j = json.loads(open(f).read().strip().split("\n")[-1])
        print(c, "value", round(j["value"]), "ms/step", round(j["ms_per_step"], 3), "kernel-only", round(j["kernel_only"]["value"]), "frac", round(j["roofline"]["frac"], 4),
              "kernel_ms", round(j["roofline"]["kernel_ms"], 3), "cpu", [round(x["value"], 1) for x in j.get("cpu_baselines", [])],
              "hbm_leg", j.get("hbm_resident", {}).get("roofline", {}).get("frac"), "impact", j.get("impact_stream", {}).get("frac"))
if os.path.exists(f"{O}/stats/cfg5_kernel_stats.csv"):
    open(f"{P}/final_cfg5_kernel_stats.csv", "w").write(re.sub(r'\(ns::[^"]*\)"', '"', open(f"{O}/stats/cfg5_kernel_stats.csv").read()))
    open(f"{P}/final_cfg5_kernel_trace_head.csv", "w").writelines(open(f"{O}/stats/cfg5_kernel_trace.csv").readlines()[:40])
    print(open(f"{P}/final_cfg5_kernel_stats.csv").read().split("\n")[1][:200])
for a, b in (("law_bench.txt", "final_law_bench.txt"), ("law_big20.txt", "final_law_bench_big20.txt"), ("e2e.txt", "final_e2e_host_inclusive.txt"), ("tests.txt", "final_gpu_tests.txt"),
             ("n2_strong.json", "final_n2_gloo_rehearsal_strong.json"), ("n2_weak.json", "final_n2_gloo_rehearsal_weak.json"),
             ("facade_bench.txt", "final_facade_bench.txt")):
    if os.path.exists(f"{O}/{a}"):
        shutil.copy(f"{O}/{a}", f"{P}/{b}")
if os.path.exists(f"{O}/pmc_fetch/cfg5_counter_collection.csv"):
    f = counters(f"{O}/pmc_fetch/cfg5_counter_collection.csv", "FETCH_SIZE"); w = counters(f"{O}/pmc_write/cfg5_counter_collection.csv", "WRITE_SIZE")
    fm = sum(x[2] for x in f) / len(f); wm = sum(x[2] for x in w) / len(w)
    # a sharing batch (ns_ctx_share_scores) runs k_share_scores in front of every scoring launch: one step = both kernels
    fs = counters(f"{O}/pmc_fetch/cfg5_counter_collection.csv", "FETCH_SIZE", "k_share_scores"); ws = counters(f"{O}/pmc_write/cfg5_counter_collection.csv", "WRITE_SIZE", "k_share_scores")
    if fs:
        print("k_share_scores per launch: FETCH_SIZE", sum(x[2] for x in fs) / len(fs), "KB, WRITE_SIZE", sum(x[2] for x in ws) / max(len(ws), 1), "KB")
        fm += sum(x[2] for x in fs) / len(fs); wm += sum(x[2] for x in ws) / max(len(ws), 1)
        f += fs; w += ws
    traffic = (2 * fm + wm) * 1024
    print("cfg5 L2-miss traffic per launch (2 x FETCH + WRITE):", traffic)
    with open(f"{P}/final_cfg5_pmc_traffic.csv", "w") as out:
        out.write("counter,dispatch_id,kernel,value_KB\n")
        for name, rows in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
            for d, k, v in rows:
                out.write(f"{name},{d},{k.split('(')[0][:60].replace(',', ';')},{v}\n")
    tj["cfg5_v0_q16384"] = traffic
    rows = list(csv.DictReader(open(f"{O}/pmc_sq/cfg5_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if "uscore" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {c: sum(v) / len(v) for c, v in agg.items()}
    cyc = m["SQ_BUSY_CYCLES"] / 32; cap = cyc / 4 * 1024
    print(f"VALU {m['SQ_INSTS_VALU']:.4g} SALU {m['SQ_INSTS_SALU']:.4g} LDS {m['SQ_INSTS_LDS']:.3g} VALU issue {m['SQ_ACTIVE_INST_VALU'] / cap:.2f} SALU issue {m['SQ_INSTS_SALU'] / cap:.2f} "
          f"waves avg {m['SQ_WAVE_CYCLES'] * 4 / cyc:.0f} wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f} wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}")
    open(f"{P}/final_cfg5_pmc_sq.csv", "w").write("counter,mean_per_dispatch\n" + "\n".join(f"{k},{v:.6g}" for k, v in m.items()) + "\n")
# the hbm_resident leg of bench.py alone (bench.py --hbm-only): kernel stats and L2-miss traffic of exactly that launch
if os.path.exists(f"{O}/hbm_stats/hbm_leg_kernel_stats.csv"):
    open(f"{P}/final_hbm_leg_kernel_stats.csv", "w").write(re.sub(r'\(ns::[^"]*\)"', '"', open(f"{O}/hbm_stats/hbm_leg_kernel_stats.csv").read()))
    print("hbm leg:", open(f"{P}/final_hbm_leg_kernel_stats.csv").read().split("\n")[1][:200])
    if os.path.exists(f"{O}/hbm_stats/hbm_leg.json") and os.path.getsize(f"{O}/hbm_stats/hbm_leg.json") > 10:
        shutil.copy(f"{O}/hbm_stats/hbm_leg.json", f"{P}/final_hbm_leg_under_rocprof.json")
if os.path.exists(f"{O}/hbm_fetch/hbm_leg_counter_collection.csv"):
    f = [x for x in counters(f"{O}/hbm_fetch/hbm_leg_counter_collection.csv", "FETCH_SIZE") if x[2] > 1000]
    w = [x for x in counters(f"{O}/hbm_write/hbm_leg_counter_collection.csv", "WRITE_SIZE")] if os.path.exists(f"{O}/hbm_write/hbm_leg_counter_collection.csv") else []
    fm = sum(x[2] for x in f) / len(f); wm = (sum(x[2] for x in w) / len(w)) if w else 0.0
    traffic = (2 * fm + wm) * 1024
    print("hbm leg L2-miss traffic per launch (2 x FETCH + WRITE):", traffic, "launches", len(f))
    with open(f"{P}/final_hbm_leg_pmc_fetch.csv", "w") as out:
        out.write("counter,dispatch_id,kernel,value_KB\n")
        for name, rows in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
            for d, k, v in rows:
                out.write(f"{name},{d},{k.split('(')[0][:60].replace(',', ';')},{v}\n")
    tj["cfg5_big20_q2048"] = traffic
# big-index FETCH_SIZE per law: the laws run in the order given on the command line, 1 untimed + 3 timed launches each
laws = ["cfg5", "cfg5_thin", "cfg5_gen", "cfg5_tile", "r8"]
big = {}
for tag in ("raw", "nodeal", "pk1", "pk2"):
    f = f"{O}/big_fetch_{tag}/counter_collection.csv"
    if not os.path.exists(f):
        continue
    vals = [v for _, k, v in counters(f, "FETCH_SIZE") if v > 1000]          # drop the warm-up query of Engine.reload (tiny)
    per = len(vals) // len(laws)
    big[tag] = {l: 2 * 1024 * sum(vals[i * per + 1:(i + 1) * per]) / max(per - 1, 1) for i, l in enumerate(laws)}
if big:
    with open(f"{P}/final_big20_fetch_per_law.csv", "w") as out:
        out.write("law,stream,L2_miss_bytes_per_launch(2 x FETCH_SIZE)\n")
        for tag, d in big.items():
            for l, v in d.items():
                out.write(f"{l},{tag},{v:.0f}\n")
    print("big-index L2-miss bytes per launch:", {t: {l: round(v / 1e9, 2) for l, v in d.items()} for t, d in big.items()})
    if "nodeal" in big:
        tj["cfg5_big20_q2048_no_xcd_dealing"] = big["nodeal"]["cfg5"]
    if "pk1" in big:
        tj["cfg5_big20_q2048_packed1"] = big["pk1"]["cfg5"]
for a, b in (("invert_bench.json", "final_invert_bench.json"), ("invert_bench_1m.json", "final_invert_bench_1m.json"), ("sem_bench.json", "final_sem_bench.json")):
    if os.path.exists(f"{O}/{a}") and os.path.getsize(f"{O}/{a}") > 10:
        shutil.copy(f"{O}/{a}", f"{P}/{b}")
        j = json.loads(open(f"{O}/{a}").read().strip().split("\n")[-1])
        print(a, "value", round(j["value"]), "device_ms", round(j["device_ms"], 3), "frac", round(j["roofline"]["frac"], 4), {k: j[k] for k in ("call_s", "total_s", "ref_s") if k in j})
for a, b in (("invert_kernel_stats.csv", "final_invert_kernel_stats.csv"), ("sem_kernel_stats.csv", "final_sem_kernel_stats.csv")):
    if os.path.exists(f"{O}/{a}"):
        open(f"{P}/{b}", "w").write(re.sub(r'\(.*?\)"', '"', open(f"{O}/{a}").read()))
json.dump(tj, open(tj_path, "w"), indent=1)
