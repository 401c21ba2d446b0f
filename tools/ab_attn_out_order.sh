#!/bin/bash
# Tile order of the N = 768 residual GEMM at K = 768 (attention-out): groups of 8 row panels (default for K <= 1024) against plain
# row-major (rr_set_gemm_stagger 56) — launch time, alternating processes, and FETCH_SIZE per launch.   tools/ab_attn_out_order.sh <tag>
tag=$1
out=gpurun_out/${tag}_attn_out_tile_order.log
: > $out
for rep in 1 2 3; do
  for sg in 0 56 50 51; do
    echo "== stagger $sg rep $rep" >> $out
    python tools/gemm_epilogue_timeline.py --no-timeline --shapes attn_out --stagger $sg --rounds 5 2>/dev/null | grep product >> $out
  done
done
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
for sg in 0 56; do
  rm -rf /tmp/pmc_$sg
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$sg -o f -- python3 $root/tools/gemm_epilogue_timeline.py --no-timeline --shapes attn_out --stagger $sg --rounds 1 > /dev/null 2>&1
  python3 - <<P >> $root/$out
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pmc_$sg/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "gemm_kernel_hp<4, 1, 3" in r["Kernel_Name"]]
print("stagger $sg: FETCH_SIZE x2 per launch of gemm_kernel_hp<4,1,3> (attention-out shape): %.3f GB over %d launches" % (2*sum(v)/len(v)*1024/1e9, len(v)))
P
done
cd $root
cat $out
