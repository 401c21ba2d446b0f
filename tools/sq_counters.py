"""Aggregate two rocprofv3 --pmc passes of SQ counters over bench.py into per-kernel wave-state shares and MFMA busy.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
              SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmcG1 -o a \
              -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA \
              SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace ... -d gpurun_out/pmcG2 -o b -- python3 bench.py ...
    python tools/sq_counters.py gpurun_out/pmcG1 gpurun_out/pmcG2 > profiles/rNN_sq_counters.json

Units (MI355X_MICROARCH.md "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs.  WAIT_ANY (parked at
s_waitcnt / s_barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.  Only the big launches of the
step (>= 200 workgroups) are aggregated.  The shader clock is taken from the persistent GEMM, whose 2048 waves live for
the whole launch: clock = 4 * WAVE_CYCLES / 2048 / duration; mfma_busy = MFMA_BUSY_CYCLES / (1024 SIMDs * duration * clock).
"""
import collections
import csv
import glob
import json
import re
import sys

PAT = re.compile(r"((gemm_kernel_hp8|gemm_kernel_hp|attn_fixed64_kernel|attn_fixed_kernel|attn_fwd_kernel|layernorm_kernel)(<[^>]*>)?)")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:3]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = PAT.search(r["Kernel_Name"])
            if not m or int(r["Grid_Size"]) < 200 * 256:
                continue
            cnt[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                dur[m.group(1)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
mean = lambda v: sum(v) / len(v)
clocks = []
for k, v in cnt.items():
    if k.startswith("gemm_kernel_hp") and "SQ_WAVE_CYCLES" in v:
        clocks.append(4.0 * mean(v["SQ_WAVE_CYCLES"]) / 2048.0 / mean(dur[k]))     # cycles per ns = GHz
clock = mean(clocks) if clocks else None
out = {}
for k, v in sorted(cnt.items()):
    wc = mean(v["SQ_WAVE_CYCLES"])
    e = {"launches": len(v["SQ_WAVE_CYCLES"]), "avg_duration_ms": mean(dur[k]) / 1e6,
         "wave_state_share": {n: mean(v[c]) / wc for n, c in (("parked_waitcnt_barrier", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"),
                                                               ("issuing", "SQ_ACTIVE_INST_ANY")) if c in v}}
    if clock and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        e["mfma_busy"] = mean(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024.0 * mean(dur[k]) * clock)
    if "SQ_LDS_IDX_ACTIVE" in v and clock:
        e["lds_array_busy"] = mean(v["SQ_LDS_IDX_ACTIVE"]) / (256.0 * mean(dur[k]) * clock)
    if "SQ_LDS_BANK_CONFLICT" in v and "SQ_LDS_IDX_ACTIVE" in v:
        e["lds_bank_conflict_share_of_lds_cycles"] = mean(v["SQ_LDS_BANK_CONFLICT"]) / max(1.0, mean(v["SQ_LDS_IDX_ACTIVE"]))
    e["raw_mean"] = {c: mean(x) for c, x in sorted(v.items())}
    out[k] = e
print(json.dumps({"note": "rocprofv3 --pmc SQ counters, two passes over bench.py (c3, 800 pairs, 1 warm-up + 1 step), big launches only",
                  "shader_clock_ghz_under_load": clock, "per_kernel": out}, indent=1))
