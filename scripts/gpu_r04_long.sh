#!/bin/bash
# round 4, long validation on the final library: FAST vs STRICT over 100 000 random scenes on each sky, the per-config parity report, a 4-minute soak
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04long
mkdir -p $O
(time python scripts/fuzz_modes.py 100000 927) 2> $O/fuzz_u.time > $O/fuzz_modes_100000.json
(time python scripts/fuzz_modes.py 100000 2026 clustered) 2> $O/fuzz_c.time > $O/fuzz_modes_clustered_100000.json
(time timeout 900 python scripts/parity_report.py) 2> $O/parity_report.err | grep -v amdgpu > $O/parity_report.jsonl
(time timeout 600 python scripts/soak.py 240 77) > $O/soak.txt 2>&1
cut -c1-420 $O/fuzz_modes_100000.json; echo; cut -c1-420 $O/fuzz_modes_clustered_100000.json; echo
cut -c1-260 $O/parity_report.jsonl; tail -n 2 $O/soak.txt | cut -c1-300; tail -n 3 $O/fuzz_u.time $O/fuzz_c.time
