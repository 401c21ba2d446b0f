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


def short(name):
    """'void (anonymous namespace)::gemm_kernel_hp<1, 1, 0, 0, true>(unsigned short const*, ...' -> 'gemm_kernel_hp<1, 1, 0, 0, true>'"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def load(d, counter, key=klass):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = key(r["Kernel_Name"])
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
fk, nfk = load(fd, "FETCH_SIZE", short)
wk, nwk = load(wd, "WRITE_SIZE", short)
per_kernel = {k: {"launches": nfk[k], "read_gb_per_launch": round(2.0 * fk[k] * 1024.0 / nfk[k] / 1e9, 4),
                  "write_gb_per_launch": round(wk.get(k, 0.0) * 1024.0 / max(1, nwk.get(k, 0)) / 1e9, 4)}
              for k in sorted(fk, key=lambda k: -fk[k]) if 2.0 * fk[k] * 1024.0 > 1e8}
print(json.dumps({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py (c3, 800 pairs, "
                          "1 warm-up + 1 step); FETCH_SIZE doubled per the gfx950 correction",
                  "per_kernel_class": out, "per_kernel": per_kernel}, indent=1))
