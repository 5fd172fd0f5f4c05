#!/usr/bin/env python3
"""Instruction mix of the trace kernels per basic block, from the compiler's own assembly (hipcc cross-compiles: no GPU needed).
The stepping loops are the blocks with a hundred f64 instructions and more: FAST = FAST_LOOP_IN_LINE, the sum of `.Lbs_loop` and its join blocks
(csrc/fast_loop_asm.h, two steps per trip: 2 x (65 full-rate f64 VALU + 2 v_rsq_f64), 2 x 9 SALU + the branches; `.Lbs_slow*` = the out-of-line
stages 1 and 3 with their own v_rsq_f64 that a few steps per hundred visit), STRICT = the two big blocks of the compiled loop (178 + 4 rsq + 4 rcp per step).  These are the numbers `LOOP_VALU` in bench_legs.py carries (`roofline.valu_issue_frac`).
Usage: isa_hot_blocks.py [--strict] [--all] [extra hipcc flags ...]     (--all: every block with >= 6 VALU, not only the loops)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = [a for a in sys.argv[1:] if a not in ("--strict", "--all")]
kernel = "trace_frame_kernelILb0" if "--strict" in sys.argv else "trace_frame_kernelILb1"
with tempfile.TemporaryDirectory() as d:
    s = os.path.join(d, "k.s")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DBS_PAD_NOPS=0",
                           *flags, "--cuda-device-only", "-S", os.path.join(ROOT, "blackstar_amd/csrc/trace_kernel.hip"), "-o", s], stderr=subprocess.DEVNULL)
    text = open(s).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*" + kernel + r"\S*:", l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
rows, cur = [], None
for l in text[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+|\.Lbs_\w+):", l)
    if m:
        cur = dict(name=m.group(1).replace("; %", ""), valu=0, f64=0, trans=0, mov=0, salu=0, branch=0, lds=0, vmem=0, depth="")
        d = re.search(r"Depth=(\d)", l)
        cur["depth"] = d.group(1) if d else ""
        rows.append(cur)
        continue
    t = l.strip()
    if cur is None or not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur["valu"] += 1
        cur["f64"] += "_f64" in op
        cur["trans"] += bool(re.match(r"v_(rsq|rcp|sqrt|exp|log|sin|cos)_", op))
        cur["mov"] += op.startswith("v_mov")
    elif op.startswith("s_"):
        cur["salu"] += 1
        cur["branch"] += "branch" in op
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_")):
        cur["vmem"] += 1
print(f"{kernel}: {sum(r['valu'] for r in rows)} VALU instructions in {len(rows)} blocks")
# the assembly stepping loop is several blocks (its join labels split it): add them up -- in line = .Lbs_loop up to the first out-of-line block
names = [r["name"] for r in rows]
lo = next((i for i, n in enumerate(names) if n.startswith(".Lbs_loop")), None)
if lo is not None:
    hi = next((i for i in range(lo, len(rows)) if names[i].startswith((".Lbs_slow", ".Lbs_cross"))), len(rows))
    tot = {k: sum(r[k] for r in rows[lo:hi]) for k in ("valu", "f64", "trans", "mov", "salu", "branch", "lds", "vmem")}
    rows.append(dict(name="FAST_LOOP_IN_LINE", depth="", **tot))
for r in rows:
    if r["f64"] >= 100 or r["name"].startswith(".Lbs_slow") or ("--all" in sys.argv and r["valu"] >= 6):
        print(f"{r['name']:16s} depth {r['depth'] or '-'}  VALU {r['valu']:3d} (f64 {r['f64']:3d}, of them transcendental {r['trans']:2d}; v_mov {r['mov']})  "
              f"SALU {r['salu']:3d} (branches {r['branch']})  LDS {r['lds']}  VMEM {r['vmem']}")
