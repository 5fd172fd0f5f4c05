#!/bin/bash
# round 4: the GPU suite as the driver runs it, then the partition trial A/B (scripts/partition_trial_ab.py) -- run on a second box
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04e}
mkdir -p $O
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
(time timeout 1500 python scripts/partition_trial_ab.py 36) 2> $O/partition_ab.err > $O/partition_trial_ab.jsonl
tail -n 6 $O/pytest_gpu.log | cut -c1-200
python - <<PY
import json
for ln in open("$O/partition_trial_ab.jsonl"):
    d = json.loads(ln)
    if "scene" in d:
        print(d["scene"], d["frame"], d["form"], d["mode"], d["bloom"], d["divider"], "trial", d["trial_ms"], "calls", d.get("trial_calls"), "->", d["choice"], "forced", d["forced"], "auto", d["auto"], "regret", d["regret_pct"], d["agrees"])
    else:
        print(d)
PY
