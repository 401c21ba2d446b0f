"""Host-side study of the device's rounding points (no GPU needed).

Emulates, on top of the fp32 oracle, where the HIP path rounds to bf16
(GEMM operands, stored Q/K/V, softmax probabilities) while keeping fp32
accumulation, fp32 residual stream, fp32 LayerNorm/softmax/GELU — and reports the
logit drift against the pure-fp32 oracle.  Used to choose the storage precision of
each intermediate before writing kernels, and to justify the tolerance in
tests/ (see DESIGN.md "Numerics").

    python tools/precision_study.py --case c1
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rerank_oracle as O  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def mm_bf16(x, W):
    return bf(x) @ bf(W).t()


def mha_bf16(q, k, v, heads, add_mask):
    B, Tq, H = q.shape
    Tk = k.shape[1]
    dh = H // heads
    q, k, v = bf(q), bf(k), bf(v)                       # stored bf16 by the QKV epilogue
    qh = q.view(B, Tq, heads, dh).transpose(1, 2)
    kh = k.view(B, Tk, heads, dh).transpose(1, 2)
    vh = v.view(B, Tk, heads, dh).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if add_mask is not None:
        s = s + add_mask
    p = bf(torch.softmax(s, -1))                        # P -> bf16 for the PV MFMA
    return (p @ vh).transpose(1, 2).reshape(B, Tq, H)


def autocast_forward(cfg, w, ids, am, tt, Bq, K, img):
    """Lightning bf16-mixed equivalent: torch.autocast over the fp32 oracle."""
    with torch.autocast("cpu", dtype=torch.bfloat16):
        return O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="c1")
    a = ap.parse_args()
    shapes = {"c1": (2, 5, 128, False), "c2": (1, 20, 256, False), "c3s": (1, 4, 512, True),
              "c1mm": (2, 5, 128, True)}
    Bq, K, S, vision = shapes[a.case]
    cfg = O.OracleConfig()
    for hf_init in (False, True):
        w = O.make_weights(cfg, 0, vision, hf_init=hf_init)
        ids, am, tt = O.make_pair_batch(cfg, Bq, K, S)
        img = O.make_image_feats(cfg, Bq) if vision else (None, None)
        with torch.no_grad():
            ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1]).logits
            keep = O.multi_head_attention
            O.multi_head_attention = mha_bf16
            dev = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1], mm=mm_bf16).logits
            O.multi_head_attention = keep
            ac = autocast_forward(cfg, w, ids, am, tt, Bq, K, img).logits.float()
        print(f"case {a.case} hf_init={hf_init}: |logit| mean {ref.abs().mean():.4f} max {ref.abs().max():.4f}")
        print(f"  device-emulation vs fp32 : max {(dev - ref).abs().max():.3e} mean {(dev - ref).abs().mean():.3e}")
        print(f"  torch autocast   vs fp32 : max {(ac - ref).abs().max():.3e} mean {(ac - ref).abs().mean():.3e}")
        print(f"  device-emulation vs autocast: max {(dev - ac).abs().max():.3e}")
