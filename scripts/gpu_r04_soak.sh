#!/bin/bash
# round 4: soak of the batch paths incl. same-shape rounds that run, continue and end partition trials; wall time of the default bench command
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $O
(time timeout 600 python scripts/soak.py ${1:-150} 31) > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/soak.txt
(time python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_time.txt
tail -n 8 $O/soak.txt | cut -c1-300; cat $O/bench_time.txt
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value'],1), d['valid'], {k:(round(v['Mpixel_s'],1), v.get('frames_identical', v.get('identical_to_one_device'))) for k,v in d['with_d2h'].items()})"
