#!/usr/bin/env python3
"""Static instruction census of one kernel's assembly (hipcc -S -gline-tables-only), basic block by basic block.

    python3 tools/dbg/asm_blocks.py kernel.s [first_asm_line last_asm_line]

Per block: asm line, label, VALU / SALU / LDS / VMEM / waitcnt+nop counts and the source lines (file:line span per file) its
instructions come from.  Diagnostic only (tools/dbg): used to price the sections of the scoring bodies against the event counts
of the counting build (tools/dbg/count_run.py).
"""
import re, sys, collections
path = sys.argv[1]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
files = {}
cur = (0, 0)
blk = None
blocks = []
def new(label, ln):
    global blk
    blk = {"label": label, "at": ln, "v": 0, "s": 0, "lds": 0, "vmem": 0, "wait": 0, "br": 0, "src": collections.defaultdict(list)}
    blocks.append(blk)
new("<entry>", 0)
for i, ln in enumerate(open(path), 1):
    t = ln.strip()
    m = re.match(r"\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]*)\"", t)
    if m: files[int(m.group(1))] = m.group(2); continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r"(\.LBB[0-9_]+):", t)
    if m: new(m.group(1), i); continue
    if not t or t[0] in ".;#" or t.endswith(":"): continue
    op = t.split()[0]
    if i < lo or i > hi: continue
    if op in ("s_waitcnt", "s_nop"): blk["wait"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch": blk["br"] += 1; blk["s"] += 1
    elif op.startswith("v_"): blk["v"] += 1
    elif op.startswith("s_"): blk["s"] += 1
    elif op.startswith("ds_"): blk["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): blk["vmem"] += 1
    else: continue
    blk["src"][cur[0]].append(cur[1])
tot = collections.Counter()
for b in blocks:
    n = b["v"] + b["s"] + b["lds"] + b["vmem"]
    if n == 0: continue
    src = " ".join(f"{f}:{min(l)}-{max(l)}" for f, l in sorted(b["src"].items()))
    print(f"{b['at']:>6} {b['label']:<14} V {b['v']:>3} S {b['s']:>3} L {b['lds']:>2} M {b['vmem']:>2} w {b['wait']:>2}   {src}")
    for k in ("v", "s", "lds", "vmem"): tot[k] += b[k]
print("total", dict(tot))
