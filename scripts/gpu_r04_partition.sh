set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu -k "partition or png_batch_at_full or rgb8_batch" -rs -s) > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.log
(time timeout 1200 python scripts/partition_trial_ab.py 36) 2> $O/partition_ab.err > $O/partition_trial_ab.jsonl
tail -n 30 $O/pytest_sel.log | cut -c1-250
python - <<PY
import json
for ln in open("$O/partition_trial_ab.jsonl"):
    d = json.loads(ln)
    if "scene" in d:
        print(d["scene"], d["frame"], d["form"], d["mode"], "bloom", d["bloom"], d["divider"], "trial", d["trial_ms"], "->", d["choice"], "forced", d["forced"], "auto", d["auto"], "regret", d["regret_pct"], d["agrees"])
    else:
        print(d)
PY
