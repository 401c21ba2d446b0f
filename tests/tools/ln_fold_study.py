"""Host-side study (no GPU): what does folding LayerNorm into the consumer GEMM cost in logit drift?

Device plan (DESIGN.md "LayerNorm folded into the consumer GEMM"): the producer epilogue writes the pre-LayerNorm row x
(fp32 for the residual, 16-bit as the next MFMA operand) and per-row (sum, sum of squares); the consumer computes
  LN(x) W^T + b = rstd * (x16 W'^T - mu * c) + d,   W' = 16bit(W * gamma),  c_n = sum_k W'_nk,  d_n = sum_k beta_k W_nk + b_n
so the operand rounding moves from LN(x) to x.  This script emulates exactly that on the fp32 oracle for the QKV and
FFN-up GEMMs of every layer but the first of each stack, next to the current rounding points (device_rounding) and the
fp32 forward, on the committed goldens."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helpers import O, golden_inputs, load_golden  # noqa: E402


def folded_linear(x, gamma, beta, eps, W, b, rd):
    """rstd * (rd(x) rd(W*gamma)^T - mu * c) + d with one-pass statistics from (sum, sum of squares), as the device forms them."""
    n = x.shape[-1]
    s1, s2 = x.sum(-1, keepdim=True), (x * x).sum(-1, keepdim=True)
    mu = s1 / n
    var = (s2 / n - mu * mu).clamp_min(0.0)
    rstd = torch.rsqrt(var + eps)
    Wp = rd(W * gamma[None, :])
    c = Wp.sum(1)
    d = (W.double() @ beta.double()).float() + (b if b is not None else 0.0)
    return rstd * (rd(x) @ Wp.t() - mu * c) + d


def layer_folded(pre_in, g_in, b_in, w, p, heads, eps, add_mask, rd, mha):
    """One post-LN BertLayer whose input is the PREVIOUS LayerNorm's input `pre_in` (+ that LN's gamma/beta); returns
    this layer's LN2 input and LN2's parameters.  Residuals use the exact fp32 LayerNorm (as the device's ln_apply)."""
    h = F.layer_norm(pre_in, (pre_in.shape[-1],), g_in, b_in, eps)            # fp32 residual value
    lin = lambda nm: folded_linear(pre_in, g_in, b_in, eps, w[p + nm + ".weight"], w.get(p + nm + ".bias"), rd)
    q, k, v = lin(".attention.self.query"), lin(".attention.self.key"), lin(".attention.self.value")
    ctx = mha(q, k, v, heads, add_mask)
    pre1 = rd(ctx) @ rd(w[p + ".attention.output.dense.weight"]).t() + w[p + ".attention.output.dense.bias"] + h
    g1, b1 = w[p + ".attention.output.LayerNorm.weight"], w[p + ".attention.output.LayerNorm.bias"]
    a = F.layer_norm(pre1, (pre1.shape[-1],), g1, b1, eps)
    inter = O.gelu_erf(folded_linear(pre1, g1, b1, eps, w[p + ".intermediate.dense.weight"], w[p + ".intermediate.dense.bias"], rd))
    pre2 = rd(inter) @ rd(w[p + ".output.dense.weight"]).t() + w[p + ".output.dense.bias"] + a
    return pre2, w[p + ".output.LayerNorm.weight"], w[p + ".output.LayerNorm.bias"]


def forward_folded(cfg, w, ids, am, tt, Bq, K, dtype):
    rd = lambda t: t.to(dtype).float()
    with O.device_rounding(dtype) as mm:
        mha = O._MHA[-1]
        p = "context_text_encoder.bert_model"
        h = O.bert_embeddings(w, p + ".embeddings", cfg.ln_eps, ids, tt)
        mask = O.extended_mask(am)
        h = O.bert_layer(h, w, f"{p}.encoder.layer.0", cfg.heads, cfg.ln_eps, mask, mm=mm)      # layer 0: LN'd operand as now
        # re-derive layer 0's LN2 input to continue in folded form: run it again keeping pre2 (cheap at these sizes)
        pre, g, b = layer0_pre(cfg, w, p, ids, tt, mask, mm)
        for i in range(1, cfg.layers):
            pre, g, b = layer_folded(pre, g, b, w, f"{p}.encoder.layer.{i}", cfg.heads, cfg.ln_eps, mask, rd, mha)
        hs = F.layer_norm(pre, (pre.shape[-1],), g, b, cfg.ln_eps)
        text = mm(hs, w["context_text_encoder_linear.weight"]) * O.token_mask(ids)[..., None]
        Q = F.normalize(text, p=2, dim=2)
        x = O.linear(Q, w, "cross_encoder_input_mapping", mm)
        l1, l2 = O.cross_encoder(cfg, w, x, O.token_mask(ids), None, mm, None)
    return l1.reshape(-1)


def layer0_pre(cfg, w, p, ids, tt, mask, mm):
    h = O.bert_embeddings(w, p + ".embeddings", cfg.ln_eps, ids, tt)
    pl = f"{p}.encoder.layer.0"
    q, k, v = (O.linear(h, w, pl + ".attention.self." + n, mm) for n in ("query", "key", "value"))
    ctx = O._MHA[-1](q, k, v, cfg.heads, mask)
    a = O.layer_norm(O.linear(ctx, w, pl + ".attention.output.dense", mm) + h, w, pl + ".attention.output.LayerNorm", cfg.ln_eps)
    inter = O.gelu_erf(O.linear(a, w, pl + ".intermediate.dense", mm))
    pre2 = O.linear(inter, w, pl + ".output.dense", mm) + a
    return pre2, w[pl + ".output.LayerNorm.weight"], w[pl + ".output.LayerNorm.bias"]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="c1,c2")
    a = ap.parse_args()
    torch.set_num_threads(8)
    for name in a.cases.split(","):
        g = load_golden(name)
        cfg = g["cfg"]
        assert not g["vision"]
        for hf_init in (False, True):
            w = O.make_weights(cfg, 0, False, hf_init=hf_init)
            ids, am, tt, _ = golden_inputs(g)
            with torch.no_grad():
                ref = O.full_context_forward(cfg, w, ids, am, tt, g["Bq"], g["K"]).logits.reshape(-1)
                for dt in (torch.bfloat16, torch.float16):
                    with O.device_rounding(dt) as mm:
                        cur = O.full_context_forward(cfg, w, ids, am, tt, g["Bq"], g["K"], mm=mm).logits.reshape(-1)
                    fold = forward_folded(cfg, w, ids, am, tt, g["Bq"], g["K"], dt)
                    print(f"{name} hf_init={hf_init} {str(dt)[6:]}: current rounding points {(cur - ref).abs().max():.2e}   "
                          f"LN folded into QKV/FFN-up {(fold - ref).abs().max():.2e}   (|logit| max {ref.abs().max():.3f})")
