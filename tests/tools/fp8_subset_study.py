"""VERDICT r4 item 1(c), on the DEVICE: which subset of the e4m3 configuration of BASELINE configs[4] (bert-large) ranks like the
fp32 reference?  For every committed ranking fixture of the c5 family (tests/golden/c5_sep*.npz: fp32 stock-HF logits + the
reference's own bf16-autocast logits of the same lists) and every configuration

    fp8_first_layer k   (text-encoder layers below k keep 16-bit operands; 24 = no e4m3 GEMM at all)
  x fp8_qkv 0 | 1       (0: only the FFN of an e4m3 layer takes e4m3 operands)
  x fp8_ffn_down 0 | 1  (1: FFN-down on the e4m3 ring too)

the script prints / records |dlogit| (max and centred per list), rank correlation, top-5 overlap, whether the top-5 SET is kept,
next to what the reference's autocast arithmetic does on the same list, and says per fixture whether the rule binds
(autocast keeps the top-5 with max |d| <= gap / 4).  Test infrastructure (imports the oracle for the seeded weights / inputs).

    python tests/tools/fp8_subset_study.py [--json gpurun_out/fp8_subset_study.json] [--fixtures c5_sep,c5_sep_wide,...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rmr_amd  # noqa: E402
from helpers import GOLDEN, arch_from_cfg, load_fullsize, margin_stats  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "fp8_subset_study.json"))
ap.add_argument("--fixtures", default="c5_sep,c5_sep_wide,c5_sep_g20,c5_sep_g15")
ap.add_argument("--first-layers", default="0,6,12,16,18,20,22,23,24")
a = ap.parse_args()
ks = [int(v) for v in a.first_layers.split(",")]
CONFIGS = [dict(fp8_first_layer=k, fp8_qkv=1, fp8_ffn_down=0) for k in ks]
CONFIGS += [dict(fp8_first_layer=k, fp8_qkv=0, fp8_ffn_down=0) for k in ks if k < 24]
CONFIGS += [dict(fp8_first_layer=k, fp8_qkv=1, fp8_ffn_down=1) for k in (0, 12, 18)]
out = {}


def top5(t):
    return set(t.argsort(descending=True, stable=True)[:5].tolist())


for name in a.fixtures.split(","):
    if not os.path.exists(os.path.join(GOLDEN, f"{name}.npz")):
        print(f"[{name}] fixture not present, skipped")
        continue
    cfg, w, vision, qs = load_fullsize(name)
    arch = arch_from_cfg(cfg, vision, "fp16")
    arch["fp8"] = 1
    eng = rmr_amd.RerankEngine(arch)
    eng.load_state_dict(w)
    lists = []
    for qi, q in enumerate(qs):
        sel = torch.from_numpy(q["selected"].astype(np.int64))
        ref, ac = q["fp32"][sel], q["autocast"][sel]
        gap = float(q["gap_5_6"])
        st = margin_stats(ac, ref)
        binds = bool(top5(ref) == top5(ac) and st["max_abs"] <= gap / 4)
        print(f"[{name} q{qi}] gap {gap:.3f}  logit std {ref.std():.3f} | reference bf16-autocast: |d| {st['max_abs']:.3e} centred {st['centred']:.3e} "
              f"rho {st['rho']:.4f} top-5 {st['top5']} -> the fp8 rule {'BINDS' if binds else 'does not bind'} here", flush=True)
        out[f"{name}/q{qi}/reference_autocast"] = dict(gap_5_6=gap, rule_binds=binds, top5_set_kept=top5(ref) == top5(ac), **st)
        lists.append((sel, ref, gap, binds))
    for c in CONFIGS:
        for k, v in c.items():
            eng.set_option(k, v)
        tag = f"first{c['fp8_first_layer']}_qkv{c['fp8_qkv']}_down{c['fp8_ffn_down']}"
        line = []
        for qi, (q, (sel, ref, gap, binds)) in enumerate(zip(qs, lists)):
            r = eng.forward_ids(q["ids"][sel].cuda(), q["am"][sel].cuda(), q["tt"][sel].cuda(), 1, len(sel), want_order=True)
            torch.cuda.synchronize()
            lg = r["logits"].cpu()
            st = margin_stats(lg, ref)
            kept = top5(ref) == top5(lg)
            out[f"{name}/q{qi}/{tag}"] = dict(gap_5_6=gap, top5_set_kept=bool(kept), rule_binds=binds, **c, **st)
            line.append(f"q{qi}: |d| {st['max_abs']:.3f} c {st['centred']:.3f} rho {st['rho']:.3f} {st['top5']} {'kept' if kept else 'LOST'}")
        print(f"[{name}] {tag:24s} " + "   ".join(line), flush=True)
    del eng
    torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.json), exist_ok=True)
    with open(a.json, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
print("written", a.json)
