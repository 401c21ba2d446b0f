"""GPU parity of the CLIP vision tower (rr_encode_image) — the once-per-query producer of the image features the
rerankers consume (rerank_model.py:408-411,424-426) — against stock-HF goldens and the same-rounding oracle, and the
pixels -> logits chain through the module interface."""
import ast
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, O, arch_from_cfg

pytestmark = pytest.mark.gpu

TINY_TEXT = dict(vocab_size=2000, hidden=128, layers=1, heads=2, intermediate=256, max_pos=64, ce_hidden=128, ce_heads=2,
                 ce_intermediate=256, ce_layers=1, ce_max_pos=160, li_dim=64, prefix_len=4, cross_attn_len=32)


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    kw = dict(TINY_TEXT)
    kw.update(ast.literal_eval(str(g["cfg_json"])))
    g["cfg"], g["B"] = O.OracleConfig(**kw), int(g["B"])
    return g


def _arch(cfg, dtype):
    a = arch_from_cfg(cfg, True, dtype)
    a.update(vit_layers=cfg.vit_layers, vit_heads=cfg.vit_heads, vit_intermediate=cfg.vit_intermediate,
             vit_image_size=cfg.vit_image_size, vit_patch_size=cfg.vit_patch_size)
    return a


def _engine(cfg, dtype, weight_seed=5):
    import rmr_amd
    w = O.make_weights(cfg, seed=0, vision=True)
    w.update(O.make_vit_weights(cfg, seed=weight_seed))
    eng = rmr_amd.RerankEngine(_arch(cfg, dtype))
    unexpected = eng.load_state_dict(w)
    assert unexpected == []
    return eng, w


@pytest.mark.parametrize("name", ["vit_tiny", "vit_p14", "vit_b32"])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_vision_tower_matches_golden_and_oracle(name, dtype):
    g = _load(name)
    cfg, B = g["cfg"], g["B"]
    eng, w = _engine(cfg, dtype, int(g["weight_seed"]))
    px = O.make_pixel_values(cfg, B, seed=int(g["pixel_seed"]))
    assert abs(px.double().sum().item() - float(g["pixel_checksum"])) < 1e-6       # same pixels as the generator saw
    cls, pat = eng.encode_image(px.cuda())
    torch.cuda.synchronize()
    cls, pat = cls.cpu(), pat.cpu()
    gc, gp = torch.from_numpy(g["image_cls"]), torch.from_numpy(g["image_patches"])
    with torch.no_grad(), O.device_rounding(torch.bfloat16 if dtype == "bf16" else torch.float16) as mm:
        ec, ep = O.clip_vision_forward(cfg, w, px, mm=mm)
    scale = max(gc.abs().max().item(), gp.abs().max().item())
    d32 = max((cls - gc).abs().max().item(), (pat - gp).abs().max().item())
    demu = max((cls - ec).abs().max().item(), (pat - ep).abs().max().item())
    print(f"[{name}/{dtype}] max|d| vs fp32 golden {d32:.2e}, vs same-rounding oracle {demu:.2e} (|x|max {scale:.2f})")
    assert torch.isfinite(cls).all() and torch.isfinite(pat).all()
    # un-normalised residual stream (|x| up to several units): tolerances are relative to the stream's magnitude
    tol32, tolemu = (2e-3, 1e-3) if dtype == "fp16" else (1.5e-2, 6e-3)
    assert d32 <= tol32 * max(1.0, scale) and demu <= tolemu * max(1.0, scale)


def test_pixels_to_logits_through_module():
    """FullContextRerankModel(query_pixel_values=...) with vision_encoder=True runs CLIP tower -> rerank forward inside
    the library; equals feeding the tower's outputs as image features, and tracks the fp32 oracle chain."""
    import rmr_amd
    g = _load("vit_tiny")
    cfg = g["cfg"]
    Bq, K, S = 2, 3, 48
    w = O.make_weights(cfg, seed=0, vision=True)
    w.update(O.make_vit_weights(cfg, seed=5))
    m = rmr_amd.FullContextRerankModel(dict(arch=_arch(cfg, "fp16"), loss_fn="BCE"), state_dict=w)
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=4)
    px = O.make_pixel_values(cfg, Bq, seed=9)
    cls, pat = m.engine.encode_image(px.cuda())
    out = m.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), K - 1, cls, pat)
    with torch.no_grad():
        oc, op = O.clip_vision_forward(cfg, w, px)
        ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, oc, op)
    d = (out.logits.cpu().reshape(-1) - ref.logits.reshape(-1)).abs().max().item()
    print(f"pixels->logits |dlogit| vs fp32 oracle {d:.2e}")
    assert d <= 2e-3
    # [B,1,3,H,W] pixel layout of the datasets is accepted, too
    cls5, _ = m.engine.encode_image(px[:, None].cuda())
    assert torch.equal(cls5, cls)


def test_vision_errors():
    import rmr_amd
    g = _load("vit_tiny")
    cfg = g["cfg"]
    eng, _ = _engine(cfg, "bf16")
    with pytest.raises(AssertionError):
        eng.encode_image(torch.zeros(1, 3, 32, 32).cuda())
    a = _arch(cfg, "bf16")
    a["vit_layers"] = 0
    e0 = rmr_amd.RerankEngine(a)
    e0.load_state_dict(O.make_weights(cfg, seed=0, vision=True))
    with pytest.raises(NotImplementedError):
        e0.encode_image(torch.zeros(1, 3, 64, 64).cuda())
    a["vit_layers"], a["n_patches"] = 2, 15                    # n_patches must be (image/patch)^2
    with pytest.raises(NotImplementedError):
        rmr_amd.RerankEngine(a)
    a["n_patches"] = 16
    e1 = rmr_amd.RerankEngine(a)
    with pytest.raises(KeyError):                               # CLIP tensors are required once vit_layers > 0
        e1.load_state_dict(O.make_weights(cfg, seed=0, vision=True))
