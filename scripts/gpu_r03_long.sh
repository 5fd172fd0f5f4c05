#!/bin/bash
# long validation on the final library: PNG fuzz (40 000 seeded images, GPU bytes = emulation bytes) and a 4-minute soak of every pipelined path
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03long; mkdir -p $O
timeout 400 python scripts/png_fuzz.py 40000 7 > $O/png_fuzz_40000.json 2> $O/png_fuzz.err; echo "fuzz rc=$?"; cat $O/png_fuzz_40000.json
timeout 300 python scripts/soak.py 240 31 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
