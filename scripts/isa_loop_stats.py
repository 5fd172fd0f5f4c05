#!/usr/bin/env python3
"""Compile trace_kernel.hip with -save-temps and print per-basic-block instruction mixes of the RK4 loops."""
import os, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-save-temps",
                       "-c", os.path.join(ROOT, "blackstar_amd/csrc/trace_kernel.hip"), "-o", "/dev/null"], cwd=d, stderr=subprocess.DEVNULL)
S = open(os.path.join(d, "trace_kernel-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
for kname in ["trace_frame_kernelILb1E", "trace_frame_kernelILb0E"]:
    start = [i for i, l in enumerate(S) if l.startswith("_ZN2bs12_GLOBAL__N_118" + kname)][0]
    end = [i for i in range(start, len(S)) if "s_endpgm" in S[i]][0]
    body = S[start:end]
    for h in [i for i, l in enumerate(body) if "Loop Header: Depth=" in l]:
        lab = body[h].split(":")[0].strip()
        tag = "Header=" + lab.replace(".L", "")
        idx = [i for i, l in enumerate(body) if tag in l] + [h]
        br = [i for i, l in enumerate(body) if "s_cbranch" in l and lab in l]
        region = body[min(idx):max(idx + br) + 1]
        if sum("v_rsq_f64" in l for l in region) < 3 or any("Loop Header" in l and i > 0 for i, l in enumerate(body[h + 1:max(idx + br) + 1])):
            continue
        print(kname, lab)
        blocks = []
        for l in region:
            if l.startswith(".LBB") or l.startswith("; %bb"):
                blocks.append([l.strip()[:30], Counter()])
            elif l.startswith("\t") and not l.strip().startswith((";", ".")):
                if not blocks:
                    blocks.append(["<pre>", Counter()])
                blocks[-1][1][l.strip().split()[0]] += 1
        for name, c in blocks:
            dp = sum(v for k, v in c.items() if "_f64" in k)
            v32 = sum(v for k, v in c.items() if k.startswith("v_") and "_f64" not in k)
            sa = sum(v for k, v in c.items() if k.startswith("s_"))
            print(f"  {name:32s} total {sum(c.values()):4d} dp {dp:4d} rsq {c.get('v_rsq_f64_e32', 0)} rcp {c.get('v_rcp_f64_e32', 0)} valu32 {v32:3d} salu {sa:3d}")
        if "--dump" in sys.argv:
            print("\n".join(l[:100] for l in region))
