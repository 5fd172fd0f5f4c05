#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03png; mkdir -p $O
timeout 900 python scripts/partition_large_ab.py > $O/partition_large_ab.jsonl 2> $O/partition_large_ab.err; tail -2 $O/partition_large_ab.err
timeout 900 python scripts/partition_more_ab.py > $O/partition_more_ab.jsonl 2> $O/partition_more_ab.err; tail -2 $O/partition_more_ab.err
timeout 900 python scripts/png_partition_ab.py > $O/png_partition_ab.jsonl 2> $O/png_partition_ab.err; tail -2 $O/png_partition_ab.err
timeout 600 python scripts/post_partition_ab.py > $O/post_partition_ab.txt 2> $O/post_partition_ab.err; tail -2 $O/post_partition_ab.err
cat $O/partition_large_ab.jsonl $O/partition_more_ab.jsonl $O/png_partition_ab.jsonl; tail -30 $O/post_partition_ab.txt
