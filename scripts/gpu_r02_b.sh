#!/bin/bash
# round 2, bloom: parity of every sweep path, A/B timing, rocprofv3 kernel stats of the render -> bloom -> sRGB8 pipeline
set -u
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests -q -m gpu -x -k "bloom or srgb8 or rgb8 or cpp_host or animation_single" 2>&1 | tail -15) > $O/pytest_bloom.log 2>&1
cat $O/pytest_bloom.log
timeout 300 python scripts/bloom_ab.py > $O/bloom_ab.txt 2>&1; cat $O/bloom_ab.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_rgb8 -o rgb8 -- python $GRAFT_REPO_ROOT/scripts/prof_rgb8.py > $GRAFT_REPO_ROOT/$O/prof_rgb8.log 2>&1; cd $GRAFT_REPO_ROOT
tail -n 9 $O/prof_rgb8.log
find $O/prof_rgb8 -name "*kernel_stats.csv" | head -1 | xargs cat
