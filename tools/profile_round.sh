#!/bin/bash
# One profiling round on the GPU box: rocprofv3 kernel trace + the PMC passes of `python3 bench.py` (c3), each in its own run.
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
set -e
tag=$1
root=$(pwd)
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python3 $root/bench.py $BENCH_ARGS"     # BENCH_ARGS="--workload c5 --fp8" profiles another workload (tag it accordingly)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-alt-dtype > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_kt.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_f -o f -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-e2e --no-alt-dtype > /dev/null 2> $out/${tag}_pmc_f.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_w -o w -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-e2e --no-alt-dtype > /dev/null 2> $out/${tag}_pmc_w.err
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/${tag}_sq1 -o a -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-e2e --no-alt-dtype > /dev/null 2> $out/${tag}_sq1.err
echo "sq1 done"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/${tag}_sq2 -o b -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-e2e --no-alt-dtype > /dev/null 2> $out/${tag}_sq2.err
echo "sq2 done"
cd $root
python3 tools/pmc_traffic.py $out/${tag}_pmc_f $out/${tag}_pmc_w > $out/${tag}_hbm_traffic.json
python3 tools/sq_counters.py $out/${tag}_sq1 $out/${tag}_sq2 > $out/${tag}_sq_counters.json
cp $(find $out/${tag}_kt -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# the raw per-dispatch CSVs are large: keep the summaries only
rm -rf $out/${tag}_kt $out/${tag}_pmc_f $out/${tag}_pmc_w $out/${tag}_sq1 $out/${tag}_sq2
echo "summaries written"
