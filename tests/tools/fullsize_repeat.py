"""Repeat the full-size ring-vs-simple forward comparison (tests/test_gpu_fullsize.py) to look for run-to-run variation."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import O, arch_from_cfg  # noqa: E402
import rmr_amd  # noqa: E402
from rmr_amd import _lib  # noqa: E402

lib = _lib.load()
cfg = O.OracleConfig()
w = O.make_weights(cfg, seed=0, vision=True)
eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, True, "bf16"))
eng.load_state_dict(w)
Bq, K, S = 2, 100, 512
ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=11, regime="realistic")
img = O.make_image_feats(cfg, Bq, seed=11)
args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K, img[0].cuda(), img[1].cuda(), None)
if len(sys.argv) > 2:
    assert lib.rr_set_tuning(b"ln_lite", int(sys.argv[2])) == 0
ref_a = ref_b = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    lib.rr_set_gemm_variant(-1)
    a = eng.forward_ids(*args)["logits"].clone()
    lib.rr_set_gemm_variant(0)
    b = eng.forward_ids(*args)["logits"].clone()
    lib.rr_set_gemm_variant(-1)
    torch.cuda.synchronize()
    if ref_a is None:
        ref_a, ref_b = a, b
    print(f"run {it}: ring-vs-simple {(a - b).abs().max().item():.3e}  ring-vs-run0 {(a - ref_a).abs().max().item():.3e} "
          f"simple-vs-run0 {(b - ref_b).abs().max().item():.3e}  finite {bool(torch.isfinite(a).all())}", flush=True)

# which stage is not reproducible under variant 0?  (debug taps of the last forward)
eng.set_debug(True)
lib.rr_set_gemm_variant(0)
taps0 = None
for it in range(6):
    eng.forward_ids(*args)
    torch.cuda.synchronize()
    N = Bq * K
    taps = {"text_hidden": eng.debug_read("text_hidden", N * S * cfg.hidden),
            "late_interaction": eng.debug_read("late_interaction", N * (S + cfg.prefix_len + cfg.n_patches) * cfg.li_dim),
            "ce_hidden": eng.debug_read("ce_hidden", N * (S + cfg.prefix_len + cfg.n_patches) * cfg.ce_hidden)}
    if taps0 is None:
        taps0 = taps
    else:
        msg = []
        for k in taps:
            d = (taps[k] - taps0[k]).abs()
            nz = (d > 0).nonzero().flatten()
            msg.append(f"{k}: {int((d > 0).sum())} differ, max {d.max().item():.2e}" + (f", first flat idx {int(nz[0])}" if len(nz) else ""))
        print(f"variant 0 run {it}: " + " | ".join(msg), flush=True)
lib.rr_set_gemm_variant(-1)
