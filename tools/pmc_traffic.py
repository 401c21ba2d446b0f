"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into per-kernel HBM traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/rNN_hbm_traffic.json

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x
-> doubled; WRITE_SIZE (KiB) is exact for 16-byte-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import sys

CLASSES = [("gemm_fp8", "gemm_kernel_hp8"), ("gemm_fp8", "gemm_kernel_f8"), ("gemm", "gemm_kernel"), ("attention", "attn_fwd"),
           ("attention", "attn_fixed"), ("attention_redo", "attn_redo"),
           ("layernorm", "layernorm_kernel"), ("layernorm", "layernorm_q8"), ("ln_finalize", "ln_finalize"),
           ("embed", "embed_ln"), ("other", "")]


def klass(name):
    for k, pat in CLASSES:
        if pat in name:
            return k
    return "other"


def load(d, counter):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = klass(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                n[k] += 1
    return tot, n


fd, wd = sys.argv[1], sys.argv[2]
f, nf = load(fd, "FETCH_SIZE")
w, nw = load(wd, "WRITE_SIZE")
out = {}
for k in f:
    launches = nf[k]
    rd = 2.0 * f[k] * 1024.0            # x2: gfx950 FETCH_SIZE correction
    wr = w.get(k, 0.0) * 1024.0
    out[k] = {"launches": launches, "read_bytes_per_launch": rd / launches, "write_bytes_per_launch": wr / max(1, nw.get(k, 0)),
              "hbm_bytes_per_launch": rd / launches + wr / max(1, nw.get(k, 0))}
print(json.dumps({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py (c3, 800 pairs, "
                          "1 warm-up + 1 step); FETCH_SIZE doubled per the gfx950 correction",
                  "per_kernel_class": out}, indent=1))
