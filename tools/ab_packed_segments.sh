#!/bin/bash
# The packed forward (lengths U[64, 512]) by segment cost: how many rows is one segment's fixed launch overhead worth?   tools/ab_packed_segments.sh <tag>
tag=$1
for rep in 1 2; do
  for lam in 0 512 1024 2048 4096; do
    python bench.py --regime realistic --packed --segment-cost-rows $lam --no-cpu-baseline --no-e2e --no-alt-dtype --no-profile 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('segment_cost_rows $lam rep $rep:', round(d['ms_per_step'],3), 'ms', d['config']['execution'][-60:])"
  done
done > gpurun_out/${tag}_packed_segment_cost.txt
cat gpurun_out/${tag}_packed_segment_cost.txt
