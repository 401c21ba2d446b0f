"""Which numerics option moves the fp16 drift of the bert-large goldens?  c5_full (K = 200, S = 512) against its fp32 stock-HF logits
under the default handle options and with one option changed at a time.  A re-draw of roundings moves max |d| by about +-10 %; an
option that costs accuracy SYSTEMATICALLY would show on both statistics (max and rms).   python tests/tools/fp16_option_drift.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rmr_amd  # noqa: E402
from helpers import arch_from_cfg, load_fullsize  # noqa: E402

for name in ("c5_full", "c3_full"):
    cfg, w, vision, qs = load_fullsize(name)
    q = qs[0]
    K = q["ids"].shape[0]
    img = q["img"]
    for dt in ("fp16",):
        for opts in ({}, {"ln_fold": 0}, {"resid_split": 0}, {"attn_fixed_ref": 0}, {"ce_cls_only": 0}, {"ln_fold": 0, "resid_split": 0, "attn_fixed_ref": 0}):
            eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, vision, dt))
            eng.load_state_dict(w)
            for k, v in opts.items():
                eng.set_option(k, v)
            r = eng.forward_ids(q["ids"].cuda(), q["am"].cuda(), q["tt"].cuda(), 1, K, None if img[0] is None else img[0].cuda(),
                                None if img[1] is None else img[1].cuda(), None)
            torch.cuda.synchronize()
            d = (r["logits"].cpu().reshape(-1) - q["fp32"]).double()
            print(f"[{name}/{dt}] {str(opts) if opts else 'default':60s} max |d| {d.abs().max():.3e}  rms {d.pow(2).mean().sqrt():.3e}  mean {d.mean():+.3e}", flush=True)
            del eng
            torch.cuda.empty_cache()
