#!/bin/bash
# Shard-shape A/B on the GPU box: one query's K = 13 / 25 / 50-pair shards with and without the 64 x 128
# tiles of the small-GEMM path, alternating processes.   tools/ab_shard_shapes.sh <tag>
tag=$1
for rep in 1 2; do
  for K in 13 25; do
    for hr in 0 1; do
      python bench.py --queries-per-gpu 1 --K $K --no-profile --no-cpu-baseline --no-e2e --no-alt-dtype --steps 30 --warmup 5 \
        --tuning gemm_small_half_rows=$hr 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$K half_rows=$hr rep=$rep ms_per_step', round(d['ms_per_step'],4))"
    done
  done
done > gpurun_out/${tag}_shard_half_rows_ab.txt
cat gpurun_out/${tag}_shard_half_rows_ab.txt
