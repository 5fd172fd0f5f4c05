#!/bin/bash
# round 2, first GPU call: smoke, the GPU parity suite, the bench line (N=1), and the self-launching N=2 forms in their
# documented smoke modes on a one-GPU box (single-process: contexts share device 0; torchrun: gloo, ranks share device 0).
set -u
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
export TMPDIR=/tmp
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 900 python -m pytest tests -q -m gpu -x --durations=15) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_n2_single.json 2> $O/bench_n2_single.err; echo "rc=$?" >> $O/bench_n2_single.err
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --gather > $O/bench_n2_single_gather.json 2> $O/bench_n2_single_gather.err; echo "rc=$?" >> $O/bench_n2_single_gather.err
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --launcher torchrun --gather > $O/bench_n2_torchrun.json 2> $O/bench_n2_torchrun.err; echo "rc=$?" >> $O/bench_n2_torchrun.err
tail -n 3 $O/smoke.log; tail -n 30 $O/pytest_gpu.log; for f in $O/bench_*.json; do echo "== $f"; cat $f; done; tail -n 3 $O/bench_*.err
