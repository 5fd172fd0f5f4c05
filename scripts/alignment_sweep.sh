#!/bin/bash
# Where the trace kernel's code lands modulo 32 bytes is worth up to 2 % (profiles/r06_code_alignment_ab.txt).  After a change of
# trace_kernel.hip / trace_device.h: build the eight 4-byte offsets (PAD=0..7 s_nop at the top of the kernel) HERE (hipcc cross-compiles),
# then on a GPU box -- `gpurun -- 'bash scripts/alignment_sweep.sh run'` -- time the C3 and C2 frames with each, interleaved, and set PAD in
# blackstar_amd/csrc/Makefile to the best.   Usage: scripts/alignment_sweep.sh build | run | clean
set -u
cd "$(dirname "$0")/.."
case "${1:-}" in
build)
  for n in 0 1 2 3 4 5 6 7; do make -s -C blackstar_amd/csrc PAD=$n OUT=../../variants_w_pad$n.so || exit 1; done ;;
run)
  for i in 1 2 3; do
    for n in 0 1 2 3 4 5 6 7; do
      for wl in default-aa default; do
        BLACKSTAR_LIB=$PWD/variants_w_pad$n.so python bench.py --workload $wl --cpu-seconds 0 --traffic static --form resident --no-boundary \
          --sustained-frames 0 --no-validate --steps 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$i PAD=$n $wl', round(d['ms_per_step'], 4), round(d['value'], 1))"
      done
    done
  done ;;
clean) rm -f variants_w_pad*.so ;;
*) echo "usage: $0 build | run | clean" >&2; exit 2 ;;
esac
