"""Logit drift of per-tensor-scaled e4m3 operands in the four big GEMMs (QKV, attention-out, FFN-up, FFN-down), 16-bit
attention and everything else as in the bf16 design — the numerics half of BASELINE configs[4] (SURVEY.md §7 item 8:
"report logit drift separately"; north_star's 1e-3 is a bf16 figure).  CPU only: the oracle forward with its matmul hook
replaced by an e4m3 emulation (dynamic amax / 448 scales, round to nearest even, fp32 accumulation): per tensor = the
arithmetic of csrc/gemm_fp8.hip + rr_op_quantize_fp8 (bit-exact against torch.float8_e4m3fn, tests/test_gpu_fp8.py); per
row / output channel and MX blocks of 32 (the block scales of v_mfma_scale_f32_16x16x128_f8f6f4) are the finer options.

    python tests/tools/fp8_drift_study.py            # bert-base and bert-large shapes, seeded HF-init weights
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rerank_oracle as O  # noqa: E402


def q8(x):
    s = x.abs().max().clamp_min(1e-30) / 448.0
    return (x / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s


def q8_rows(x):                                    # one scale per row (token / output channel): a rank-1 scale in the epilogue
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
    return (x / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s


def q8_mx(x):                                      # MX: one power-of-two scale per 32 consecutive K elements (the f8f6f4 block scale)
    sh = x.shape
    b = x.reshape(-1, sh[-1] // 32, 32)
    e = torch.floor(torch.log2(b.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -100) / 448.0)) + 1.0
    s = torch.exp2(e)
    return ((b / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s).reshape(sh)


def make_mm(quant):
    def mm(x, W):
        if min(W.shape) >= 512:                   # the four big GEMMs of every BertLayer
            return quant(O._bf(x)) @ quant(W).t() # activations arrive as 16-bit tensors on the device
        return O.mm_bf16(x, W)
    return mm


def run(name, cfg, Bq=2, K=8, S=128):
    cfg.loss_fn = "BCE"
    w = O.make_weights(cfg, seed=0, vision=False, hf_init=True)
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=21, regime="realistic")
    with torch.no_grad():
        ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K).logits.reshape(Bq, K)
        with O.device_rounding(torch.bfloat16) as mm:
            b16 = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(Bq, K)
        f8 = {}
        for label, quant in (("per tensor", q8), ("per row / output channel", q8_rows), ("MX blocks of 32", q8_mx)):
            with O.device_rounding(torch.bfloat16):
                f8[label] = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=make_mm(quant)).logits.reshape(Bq, K)
    top = lambda z: [tuple(sorted(r.argsort(descending=True)[:5].tolist())) for r in z]
    spread = (ref.max(1).values - ref.min(1).values).mean().item()
    print(f"{name}: |logit| max {ref.abs().max().item():.3f}, mean spread inside a candidate list {spread:.3f}")
    print(f"   bf16 operands   : max |dlogit| {(b16 - ref).abs().max().item():.2e}   same top-5 set: {top(b16) == top(ref)}")
    for label, z in f8.items():
        print(f"   e4m3, {label:24s}: max |dlogit| {(z - ref).abs().max().item():.2e}   same top-5 set: {top(z) == top(ref)}")


if __name__ == "__main__":
    torch.manual_seed(0)
    run("bert-base shape (12 x 768)", O.OracleConfig())
    run("bert-large shape (24 x 1024, configs[4])",
        O.OracleConfig(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16,
                       ce_intermediate=4096, ce_layers=1, ce_max_pos=512))
