"""Yardstick only (never on the product path): what the image's PyTorch-ROCm reaches for the path's
self-attention shape — torch.nn.functional.scaled_dot_product_attention at [pairs, heads, T, 64] with the additive key bias —
against this repo's attention launch (rr_op_attention_bf16: the fixed-reference form + its normally empty redo launch),
interleaved in ONE process on the same operands (cdna guide §5.4 rules 10, 24, 25).  Every SDPA backend the build offers is
timed on its own (flash / memory-efficient / math) and the default dispatch too, with and without the bias (the flash backends of
some builds refuse an additive mask); which ones ran is printed.  Reference seam: HF `BertSelfAttention` behind
/root/reference/src/models/rerank/attention_fusion.py:133-144.

    python tools/bench_vendor_attention.py [--pairs 800] [--heads 12] [--T 512] [--json out.json]

Layouts: ours reads the fused [rows, 3H] QKV rows (q pre-scaled by log2(e)/8) and writes [rows, H]; SDPA gets contiguous
[B, heads, T, 64] q/k/v (the layout it is fastest on; the transposes a [rows, 3H] producer would need are NOT charged to it).
"""
import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402,F401
from rmr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=800)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--json", default=None)
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B, heads, T = a.pairs, a.heads, a.T
H = heads * 64
fl = 4.0 * B * heads * T * T * 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B * T, 3 * H, generator=g) * 0.5
lens = torch.randint(T // 2, T + 1, (B,), generator=g)                  # a real key mask: the tail of every pair is padding
keep = torch.arange(T)[None, :] < lens[:, None]
record = dict(shape=dict(pairs=B, heads=heads, T=T, head_dim=64), torch=torch.__version__, hip=torch.version.hip, results=[])

try:
    from torch.nn.attention import SDPBackend, sdpa_kernel
    BACKENDS = [("default", None), ("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION),
                ("math", SDPBackend.MATH)]
except ImportError:                     # noqa
    sdpa_kernel, BACKENDS = None, [("default", None)]


def time_fn(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.launches)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


for dt, tdt, name in ((1, torch.float16, "fp16"), (0, torch.bfloat16, "bf16")):
    assert lib.rr_set_op_dtype(dt) == 0
    qkv32 = x.clone()
    q32 = qkv32[:, :H]
    # ours: q pre-scaled by log2(e)/sqrt(64) (folded into W_q at pack time in the forward), additive bias 0 / -1e30
    qkv_o = torch.cat([q32 * (math.log2(math.e) / 8.0), qkv32[:, H:]], 1).to(tdt).cuda()
    kb = torch.where(keep, 0.0, -1e30).float().cuda()
    out_o = torch.empty(B * T, H, dtype=tdt, device="cuda")

    def ours():
        assert lib.rr_op_attention_bf16(qkv_o.data_ptr(), qkv_o.data_ptr() + 2 * H, qkv_o.data_ptr() + 4 * H, 3 * H, 3 * H, kb.data_ptr(), B, heads,
                                        T, T, 1, out_o.data_ptr(), H, st) == 0
    for _ in range(20):
        ours()
    t_min, t_med = time_fn(ours)
    record["results"].append(dict(dtype=name, impl="ours rr_op_attention_bf16 (key bias)", ms_min=t_min, ms_median=t_med, tflops=fl / t_min / 1e9))
    print(f"{name} ours (key bias)                 : min {t_min:.4f} ms  median {t_med:.4f} ms  {fl / t_min / 1e9:6.0f} TFLOP/s", flush=True)

    qkv_v = qkv32.to(tdt).cuda().view(B, T, 3, heads, 64)
    q, k, v = (qkv_v[:, :, i].permute(0, 2, 1, 3).contiguous() for i in range(3))           # [B, heads, T, 64]
    bias = torch.where(keep, 0.0, float("-inf")).to(tdt).cuda()[:, None, None, :]            # [B, 1, 1, T] additive, broadcast
    bias_full = bias.expand(B, 1, T, T).contiguous()                                           # for back ends that need the rows
    ref = None
    for bname, backend in BACKENDS:
        for mname, mask in (("key bias [B,1,1,T]", bias), ("key bias [B,1,T,T]", bias_full), ("no mask", None)):
            def vendor():
                return F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            try:
                if backend is None:
                    o = vendor()
                    t_min, t_med = time_fn(vendor)
                else:
                    with sdpa_kernel([backend]):
                        o = vendor()
                        t_min, t_med = time_fn(vendor)
            except Exception as ex:      # noqa: BLE001 — a backend that refuses the arguments is part of the record
                msg = str(ex).splitlines()[0][:120]
                print(f"{name} sdpa {bname:13s} {mname:18s}: not available ({msg})", flush=True)
                record["results"].append(dict(dtype=name, impl=f"sdpa/{bname}", mask=mname, error=msg))
                continue
            d = None
            if mask is not None:
                mine = out_o.view(B, T, heads, 64).permute(0, 2, 1, 3).float()
                d = (mine - o.float()).abs().max().item()
            record["results"].append(dict(dtype=name, impl=f"sdpa/{bname}", mask=mname, ms_min=t_min, ms_median=t_med, tflops=fl / t_min / 1e9,
                                          max_abs_diff_vs_ours=d))
            print(f"{name} sdpa {bname:13s} {mname:18s}: min {t_min:.4f} ms  median {t_med:.4f} ms  {fl / t_min / 1e9:6.0f} TFLOP/s"
                  + (f"  | max |ours - sdpa| {d:.2e}" if d is not None else ""), flush=True)
            del o
lib.rr_set_op_dtype(0)
if a.json:
    with open(a.json, "w") as f:
        json.dump(record, f, indent=1)
