#!/usr/bin/env python3
"""Where does the FAST stepping loop land?  Picks the Makefile's PAD (N x s_nop at the top of trace_frame_kernel, 4 bytes each).

The C3 frame is 1.6-2.3 % slower when the kernel's code is shifted by 8-20 bytes (profiles/r06_code_alignment_ab.txt), and padding placed
BEHIND the stepping loop changes nothing: it is the loop.  In the builds measured (eight offsets each, three boxes) the fast ones are those
whose loop HEAD -- the target of the back edge, taken once per two steps by the wavefront the SIMD favours -- sits 4 to 16 bytes into a 32-byte
fetch window; 20, 24, 28 and 0 are the slow ones.  This script compiles trace_kernel.hip to assembly with PAD = 0 (hipcc cross-compiles, no
GPU), assembles it, finds that loop in trace_frame_kernel<true> (the first backward branch over >= 120 f64 instructions in <= 2600 bytes) and
prints the PAD that puts its head at offset 8, the middle of the fast range (checked on a second, differently laid out build: head at 28 with PAD 0 =
slow, PAD 2 / 3 / 4 = offsets 4 / 8 / 12 = fast, as measured).  That is the COMPILED loop (-DBS_ASM_LOOP=0), which is entered twice per two steps
(the back edge and a jump over the rare blocks).  The assembly loop (fast_loop_asm.h, the product) has ONE taken branch per two steps and wants
its head AT the start of a window: offsets 0 and 32 of 64 are the fastest on the C3 and C2 frames (-1.5 % / -1.0 % against the compiled loop at
its own best offset), 4 costs 1 %, 24 / 28 cost 0.3 % (profiles/r06_asm_loop_ab.txt).  Prints 0 and says why on stderr if anything
fails: a wrong PAD costs 1-2 %, never correctness.   Usage: pick_pad.py [extra hipcc flags ...]   (--show: the loop and its offsets)"""
import os
import re
import subprocess
import sys
import tempfile

WANT_COMPILED, WANT_ASM = 8, 0
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("BS_CSRC") or os.path.join(HERE, "..", "blackstar_amd", "csrc")
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def loop_head(extra):
    with tempfile.TemporaryDirectory() as d:
        s, o = os.path.join(d, "k.s"), os.path.join(d, "k.o")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DBS_PAD_NOPS=0", *extra, "--cuda-device-only", "-S",
                               os.path.join(CSRC, "trace_kernel.hip"), "-o", s], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", o], text=True).split("\n")
    start = next(k for k, l in enumerate(dis) if "trace_frame_kernelILb1" in l and l.endswith(">:"))
    end = next((k for k, l in enumerate(dis) if k > start and l.endswith(">:")), len(dis))
    ins = []
    for l in dis[start:end]:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    for k, (a, op, args) in enumerate(ins):
        if not (op.startswith("s_cbranch") or op == "s_branch"):
            continue
        try:
            simm = int(args.split()[0])
        except (ValueError, IndexError):
            continue
        if simm < 32768:
            continue
        target = a + 4 + (simm - 65536) * 4
        body = [i for i in ins if target <= i[0] <= a]
        if a - target <= 2600 and sum("_f64" in i[1] for i in body) >= 120:
            return target, a
    raise RuntimeError("no stepping loop found in trace_frame_kernel<true>")


def main():
    args = [x for x in sys.argv[1:] if x != "--show"]
    try:
        head, back = loop_head(args)
    except Exception as e:  # noqa: BLE001
        print(f"pick_pad.py: {type(e).__name__}: {e} -- PAD 0", file=sys.stderr)
        print(0)
        return
    want = WANT_COMPILED if "-DBS_ASM_LOOP=0" in args else WANT_ASM
    pad = ((want - head) % 32) // 4
    if "--show" in sys.argv:
        print(f"loop head 0x{head:x} (offset {head % 32} of 32), back edge at 0x{back:x}, {back - head + 4} bytes; PAD {pad} puts the head at offset {(head + 4 * pad) % 32}",
              file=sys.stderr)
    print(pad)


if __name__ == "__main__":
    main()
