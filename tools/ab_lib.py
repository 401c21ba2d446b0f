"""A/B two BUILDS of the library on the whole c3 forward on one box: alternating child processes (each loads one .so,
builds the engine, times the forward), so that box-to-box differences (+-2 %) drop out.

    tools/build_baseline.sh HEAD base          # the committed tree -> tools/bin/librerank_base.so
    python tools/ab_lib.py tools/bin/librerank_base.so reranking-multimodal-retrievers_amd/librerank_mi355.so [--rounds 3]

Prints per library the min / median step time and whether the logits of the two builds are bit-identical.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time, json, hashlib
sys.path.insert(0, %r)
import torch
from rmr_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import ctypes
_probe = ctypes.CDLL(_lib.LIB_PATH)                       # an older build lacks the newest diagnostic entry points: bind what it has
_lib._SIGS = {k: v for k, v in _lib._SIGS.items() if hasattr(_probe, k)}
import rmr_amd
from rmr_amd.synthetic import image_features, pair_batch
lib = _lib.load()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    assert lib.rr_set_tuning(k.encode(), int(v)) == 0, kv
wl = os.environ.get("RR_WORKLOAD", "c3")                 # c3 (default) | c5 (bert-large, K = 200, text-only) | c5fp8
shape = dict(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16, ce_intermediate=4096) if wl.startswith("c5") else {}
arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1, cross_encoder_max_position_embeddings=750, loss_fn="BCE"),
                         has_vision=int(wl == "c3"), compute_dtype=os.environ.get("RR_DTYPE", "fp16"), **shape)
if wl == "c5fp8":
    arch["fp8"] = 1
eng = rmr_amd.RerankEngine(arch)
eng.load_state_dict(rmr_amd.synthetic_state_dict(arch, 0, True))
Bq, K, S = (8, 100, 512) if wl == "c3" else (4, 200, 512)
ids, am, tt = [t.cuda() for t in pair_batch(arch["vocab_size"], Bq, K, S, regime=os.environ.get("RR_REGIME", "full"))]
cls, pat = [t.cuda() for t in image_features(Bq, 49, 768)] if wl == "c3" else (None, None)
ts = []
for r in range(int(sys.argv[2]) + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        out = eng.forward_ids(ids, am, tt, Bq, K, cls, pat, want_order=True)
    torch.cuda.synchronize()
    if r: ts.append((time.perf_counter() - t0) / 3)
print(json.dumps({"ms": [t * 1e3 for t in ts], "sha": hashlib.sha256(out["logits"].cpu().numpy().tobytes()).hexdigest()[:16]}))
""" % ROOT

if len(sys.argv) > 1 and sys.argv[1] == "--child":       # python3 tools/ab_lib.py --child LIB INNER [KEY=INT ...]: one child in THIS process
    sys.argv = ["-c"] + sys.argv[2:]                      # (the target of `rocprofv3 --pmc ... -- python3 tools/ab_lib.py --child ...`)
    exec(CHILD)
    sys.exit(0)
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--inner", type=int, default=5)
ap.add_argument("--tuning", action="append", default=[], help="KEY=INT applied in every child")
a = ap.parse_args()
res = {l: [] for l in a.libs}
sha = {}
for r in range(a.rounds):
    for l in a.libs:
        p = subprocess.run([sys.executable, "-c", CHILD, l, str(a.inner), *a.tuning], capture_output=True, text=True)
        if p.returncode != 0:
            print(p.stdout[-2000:], p.stderr[-4000:])
            sys.exit(1)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        res[l] += d["ms"]
        sha[l] = d["sha"]
        print(f"  round {r} {os.path.basename(l)}: min {min(d['ms']):.2f} ms", flush=True)
for l, t in res.items():
    t = sorted(t)
    print(f"{l}: min {t[0]:.2f} ms  med {t[len(t) // 2]:.2f} ms  -> {800 / t[0] * 1e3:.0f} pairs/s   logits sha {sha[l]}")
print("logits bit-identical across builds:", len(set(sha.values())) == 1)
