#!/bin/bash
# Every bench line DESIGN.md quotes, on ONE box, into gpurun_out/<tag>_*.json.log (copy what is to be judged into profiles/).
#   tools/bench_lines.sh <tag>
tag=$1
o=gpurun_out
X="--no-cpu-baseline --no-e2e"
python bench.py > $o/${tag}_c3_bench.json.log 2>/dev/null; echo c3 done
python bench.py --tuning resid_lo8=0 $X --no-alt-dtype > $o/${tag}_c3_lo16_bench.json.log 2>/dev/null; echo c3 with the fp16 lo half done
python bench.py --workload c4 $X > $o/${tag}_c4_bench.json.log 2>/dev/null; echo c4 done
python bench.py --workload L $X > $o/${tag}_L_bench.json.log 2>/dev/null; echo L done
python bench.py --workload c5 $X > $o/${tag}_c5_16bit_bench.json.log 2>/dev/null; echo c5 done
python bench.py --workload c5 --fp8 $X > $o/${tag}_c5_fp8_bench.json.log 2>/dev/null; echo c5 fp8 default done
python bench.py --workload c5 --fp8 --tuning fp8_first_layer=0 $X > $o/${tag}_c5_fp8_whole_stack_bench.json.log 2>/dev/null; echo c5 fp8 whole stack done
python bench.py --regime realistic $X --no-alt-dtype > $o/${tag}_realistic_padded.json.log 2>/dev/null; echo realistic done
python bench.py --regime realistic --packed $X --no-alt-dtype > $o/${tag}_realistic_packed16.json.log 2>/dev/null; echo packed done
for k in 13 25 50 100; do python bench.py --queries-per-gpu 1 --K $k --no-profile $X --no-alt-dtype > $o/${tag}_shard_K$k.json.log 2>/dev/null; done; echo shards done
python - <<P
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_*json.log')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],3), d.get('roofline',{}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
P
