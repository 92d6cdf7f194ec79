#!/bin/bash
# full suite (inversion rewrite, merge / pruning parity), inversion bench + its kernel stats + FETCH/WRITE, then I-cache counters
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_invert}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=5 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -4 $O/tests.txt
cd $R
timeout -k 10 300 python3 tools/invert_bench.py > $O/invert_bench.json 2> $O/invert_bench.err || { tail $O/invert_bench.err; exit 1; }
cat $O/invert_bench.json
timeout -k 10 300 python3 tools/invert_bench.py --docs 1000000 --no-cpu > $O/invert_bench_1m.json 2>> $O/invert_bench.err || exit 1
cat $O/invert_bench_1m.json
timeout -k 10 200 python3 tools/fuzz_widening.py 60  2>&1 | tail -3 | tee $O/fuzz_widening.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inv_stats -- python3 $R/tools/invert_bench.py --no-cpu > $O/inv_stats.log 2>&1 || { tail -5 $O/inv_stats.log; exit 1; }
cp $(find $O/inv_stats -name "*kernel_stats.csv" | head -1) $O/invert_kernel_stats.csv; rm -rf $O/inv_stats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/inv_$c -- python3 $R/tools/invert_bench.py --no-cpu --reps 1 > $O/inv_$c.log 2>&1 || { tail -5 $O/inv_$c.log; exit 1; }
  cp $(find $O/inv_$c -name "*counter_collection.csv" | head -1) $O/invert_$c.csv; rm -rf $O/inv_$c
done
python3 - <<PY
import csv, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open("$O/invert_%s.csv" % c)):
        if "k_iv" not in r["Kernel_Name"]: continue
        a = acc[r["Kernel_Name"].split("(")[0][:40]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    tot = sum(v[0] for v in acc.values())
    print(c, "sum over k_iv kernels (KB, all calls):", round(tot), {k: (round(v[0]), v[1]) for k, v in acc.items()})
PY
cat $O/invert_kernel_stats.csv | head -12
cd $R

