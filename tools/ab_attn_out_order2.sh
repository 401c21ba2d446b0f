#!/bin/bash
tag=$1
out=gpurun_out/${tag}_resid_tile_order.log
: > $out
for rep in 1 2 3 4 5; do
  for sg in 0 50 51 52 56; do
    echo "== stagger $sg rep $rep" >> $out
    python tools/gemm_epilogue_timeline.py --no-timeline --shapes attn_out,ffn2 --stagger $sg --rounds 5 2>/dev/null | grep product >> $out
  done
done
python3 - <<P >> $out
import re, collections
d=collections.defaultdict(list); sg=None
for l in open("$out"):
    m=re.match(r"== stagger (\d+)", l)
    if m: sg=int(m.group(1)); continue
    m=re.match(r"(\w+)\s+M=.*product ([\d.]+) ms", l)
    if m: d[(m.group(1), sg)].append(float(m.group(2)))
for k in sorted(d): print(k, "min %.3f med %.3f" % (min(d[k]), sorted(d[k])[len(d[k])//2]), d[k])
P
cat $out | tail -12
