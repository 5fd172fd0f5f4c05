#!/usr/bin/env python3
"""Per-basic-block instruction mix of the first FAST (or --strict) stepping loop printed by isa_loop_stats.py --dump:
shows which blocks are the per-step hot path (the big all-f64 ones) and what else sits in them."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = "STRICT" if "--strict" in sys.argv else "FAST"
out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/isa_loop_stats.py"), "--dump"], capture_output=True, text=True).stdout.split("\n")
starts = [i for i, l in enumerate(out) if l.startswith(mode + ": loop")]
ends = [i for i, l in enumerate(out) if re.match(r"^(FAST|STRICT): loop", l)] + [len(out)]
L = out[starts[0]:min(e for e in ends if e > starts[0])]
blk, cnt, order = None, {}, []
for l in L:
    m = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)
    if m:
        blk = m.group(1); cnt[blk] = dict(valu=0, f64=0, lane=0, mov=0, salu=0, ds=0, br=[]); order.append(blk); continue
    if blk and l.startswith("\t") and not l.strip().startswith((";", ".")):
        op = l.split()[0]; c = cnt[blk]
        if op.startswith("v_"):
            c["valu"] += 1
            if "_f64" in op: c["f64"] += 1
            if "lane" in op: c["lane"] += 1
            if op.startswith("v_mov"): c["mov"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if "branch" in op: c["br"].append(l.strip().split()[-1])
        elif op.startswith("ds_"): c["ds"] += 1
print(L[0])
for b in order: print(f"{b:14s}", cnt[b])
