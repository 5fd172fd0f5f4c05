#!/bin/bash
# round 5, after the evidence pass: the new lifetime test, the whole GPU suite once more, and a soak of the batch paths on the final library
# (bs_create / bs_destroy / direct_ok changed this round).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5) > $O/pytest_gpu_tail.txt
(time timeout 600 python scripts/soak.py 150 31) > $O/soak.txt 2> $O/soak.err
cat $O/pytest_gpu_tail.txt; tail -n 3 $O/soak.txt; tail -n 4 $O/soak.err
