#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
timeout 900 python scripts/partition_more_ab.py > gpurun_out/r03png/partition_more_ab.jsonl 2> gpurun_out/r03png/partition_more_ab.err; tail -2 gpurun_out/r03png/partition_more_ab.err
cat gpurun_out/r03png/partition_more_ab.jsonl
