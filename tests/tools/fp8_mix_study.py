"""Which GEMMs of a BertLayer can take e4m3 operands?  Logit drift (vs the fp32 forward) of the bert-large shape
(BASELINE configs[4]) with per-row activation scales / per-output-channel weight scales (amax / 448, the device's
layernorm_q8 + weight packing), for every subset of {QKV, attention-out, FFN-up, FFN-down}; everything else 16-bit as in
the bf16/fp16 design.  CPU only (oracle).   python tests/tools/fp8_mix_study.py [--dtype fp16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rerank_oracle as O  # noqa: E402


def q8_rows(x):
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
    return (x / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s


KINDS = {"qkv": (".attention.self.query", ".attention.self.key", ".attention.self.value"),
         "attn_out": (".attention.output.dense",), "ffn_up": (".intermediate.dense",), "ffn_down": (".output.dense",)}


def run(cfg, sets, dtype, Bq=2, K=8, S=128, hf_init=True):
    cfg.loss_fn = "BCE"
    w = O.make_weights(cfg, seed=0, vision=False, hf_init=hf_init)
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=21, regime="realistic")
    real_linear = O.linear
    with torch.no_grad():
        ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K).logits.reshape(Bq, K)
        spread = (ref.max(1).values - ref.min(1).values).mean().item()
        print(f"|logit| max {ref.abs().max().item():.3f}, mean spread inside a list {spread:.3f}, hf_init={hf_init}")
        for sel in sets:
            pats = tuple(p for k in sel for p in KINDS[k])

            def linear(x, ww, name, mm=None):
                kind = next((k for k, ps in KINDS.items() if any(name.endswith(p) for p in ps)), None)
                if kind == "ffn_down" and name.endswith(".attention.output.dense"):
                    kind = "attn_out"
                if mm is not None and ".encoder.layer." in name and kind in sel:
                    W, b = ww[name + ".weight"], ww.get(name + ".bias")
                    y = q8_rows(O._bf(x)) @ q8_rows(W).t()
                    return y if b is None else y + b
                return real_linear(x, ww, name, mm)

            O.linear = linear
            try:
                with O.device_rounding(dtype, fold=False) as mm:
                    z = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(Bq, K)
            finally:
                O.linear = real_linear
            top = lambda t: [tuple(sorted(r.argsort(descending=True)[:5].tolist())) for r in t]
            d = z - ref
            dc = d - d.mean(1, keepdim=True)               # what is left after the shift common to a query's candidates
            print(f"   e4m3 in {'+'.join(sel) if sel else '(none: 16-bit)':34s}: max |dlogit| {d.abs().max().item():.2e}  "
                  f"within-list (centred) max {dc.abs().max().item():.2e}  same top-5 sets: {top(z) == top(ref)}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    torch.set_num_threads(8)
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    large = O.OracleConfig(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16,
                           ce_intermediate=4096, ce_layers=1, ce_max_pos=512)
    sets = [(), ("qkv",), ("ffn_up",), ("qkv", "ffn_up"), ("attn_out",), ("ffn_down",), ("qkv", "ffn_up", "ffn_down"),
            ("qkv", "attn_out", "ffn_up", "ffn_down")]
    run(large, sets, dt)
