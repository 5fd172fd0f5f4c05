#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
timeout 900 python scripts/partition_large_ab.py > gpurun_out/r03png/partition_large_ab.jsonl 2> gpurun_out/r03png/partition_large_ab.err; tail -2 gpurun_out/r03png/partition_large_ab.err
cat gpurun_out/r03png/partition_large_ab.jsonl
