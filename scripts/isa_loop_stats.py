#!/usr/bin/env python3
"""Compile trace_kernel.hip with -save-temps and print the instruction mix of the RK4 stepping loops
(the innermost loops that contain >= 3 v_rsq_f64) of the two frame kernels.  --dump prints the ISA."""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-save-temps", *os.environ.get("BS_EXTRA", "").split(),
                       "-c", os.path.join(ROOT, "blackstar_amd/csrc/trace_kernel.hip"), "-o", "/dev/null"], cwd=d, stderr=subprocess.DEVNULL)
S = open(os.path.join(d, "trace_kernel-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
for kname, mode in (("trace_frame_kernelILb1E", "FAST"), ("trace_frame_kernelILb0E", "STRICT")):
    start = [i for i, l in enumerate(S) if l.startswith("_ZN2bs12_GLOBAL__N_118" + kname)][0]
    end = [i for i in range(start, len(S)) if "s_endpgm" in S[i]][0]
    body = S[start:end]
    # split into basic blocks; a block belongs to a loop if LLVM tagged it "in Loop: Header=BBx_y" (or is the header)
    blocks, cur = [], []
    for l in body:
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
            if cur:
                blocks.append(cur)
            cur = [l]
        else:
            cur.append(l)
    blocks.append(cur)
    for bi, blk in enumerate(blocks):
        head = "\n".join(blk[:3])
        if "Inner Loop Header" not in head:
            continue
        lab = blk[0].split(":")[0]
        tag = "Header=" + lab.replace(".L", "") + " "
        region = [l for k, b2 in enumerate(blocks) if k == bi or tag in "\n".join(b2[:3]) for l in b2]
        ins = [l.strip().split()[0] for l in region if l.startswith("\t") and not l.strip().startswith((";", "."))]
        c = Counter(ins)
        if c.get("v_rsq_f64_e32", 0) < 3:
            continue
        dp = sum(v for k, v in c.items() if "_f64" in k and not k.startswith("v_mov"))
        print(f"{mode}: loop {lab}: {len(ins)} instructions in the loop body region (all paths): f64 VALU {dp} (rsq {c.get('v_rsq_f64_e32', 0)}, "
              f"rcp {c.get('v_rcp_f64_e32', 0)}), v_mov_b64 {c.get('v_mov_b64_e32', 0)}, other VALU "
              f"{sum(v for k, v in c.items() if k.startswith('v_') and '_f64' not in k and k != 'v_mov_b64_e32')}, "
              f"SALU {sum(v for k, v in c.items() if k.startswith('s_'))}, LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, "
              f"scratch {sum(v for k, v in c.items() if k.startswith('scratch_'))}")
        if "--dump" in sys.argv:
            print("\n".join(l[:100] for l in region))
