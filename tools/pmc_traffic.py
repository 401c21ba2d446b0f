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


def load_by_shape(d, counter, pat):
    """Per-dispatch values of the kernel whose short name starts with `pat`, in dispatch order, split by the PARITY of the occurrence
    (the two residual GEMMs of a layer — attention-out N 768 / K 768 and FFN-down N 768 / K 3072 — are the same kernel and alternate)."""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and short(r["Kernel_Name"]).startswith(pat):
                rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows[0::2]], [v for _, v in rows[1::2]]


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
# the split-residual kernel by SHAPE.  Which parity is FFN-down is read off the data (it reads the 2.5 GB FFN
# intermediate, attention-out the 0.63 GB attention output); the same parity then labels the WRITE_SIZE pass (same dispatch order).
# (<4, 1, 11, ...>: the same epilogue with the 8-bit lo half, the default of fp16 handles since round 5; <4, 1, 3, ...>: the fp16 lo)
RESID, LO_B = "gemm_kernel_hp<4, 1, 11", 1.0
fe, fo = load_by_shape(fd, "FETCH_SIZE", RESID)
if not fe:
    RESID, LO_B = "gemm_kernel_hp<4, 1, 3", 2.0
    fe, fo = load_by_shape(fd, "FETCH_SIZE", RESID)
we, wo = load_by_shape(wd, "WRITE_SIZE", RESID)
ROWS_GB = 0.629          # 409 600 rows x 768 columns x 2 bytes
per_shape = {}
if fe and fo:
    mean = lambda v: sum(v) / max(1, len(v))
    even_is_down = mean(fe) > mean(fo)
    # algorithmic bytes: A rows (2.52 / 0.63 GB) + residual hi + lo in; hi + lo + 8-byte statistics partials out
    resid_in, rows_out = ROWS_GB * (1.0 + LO_B / 2.0), ROWS_GB * (1.0 + LO_B / 2.0) + 0.02
    for label, fr, wr_, alg_r, alg_w in (("FFN-down (N 768, K 3072)", fe if even_is_down else fo, we if even_is_down else wo, round(4 * ROWS_GB + resid_in, 2), round(rows_out, 2)),
                                         ("attention-out (N 768, K 768)", fo if even_is_down else fe, wo if even_is_down else we, round(ROWS_GB + resid_in, 2), round(rows_out, 2))):
        per_shape[RESID + ", 0, false> " + label] = {"launches": len(fr), "read_gb_per_launch": round(2.0 * mean(fr) * 1024.0 / 1e9, 4),
                                                     "write_gb_per_launch": round(mean(wr_) * 1024.0 / 1e9, 4),
                                                     "algorithmic_read_gb": alg_r, "algorithmic_write_gb": alg_w}
print(json.dumps({"per_shape": per_shape, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py (c3, 800 pairs, "
                          "1 warm-up + 1 step); FETCH_SIZE doubled per the gfx950 correction",
                  "per_kernel_class": out, "per_kernel": per_kernel}, indent=1))
