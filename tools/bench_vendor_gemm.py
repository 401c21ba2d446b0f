"""Yardstick only (never on the product path): the vendor library's GEMM (torch.mm -> hipBLASLt / rocBLAS) on the four bert-base
shapes of the path at the bench size, against this repo's persistent ring with its plain 16-bit epilogue, interleaved in ONE
process on the same random operands (cdna guide §5.4 rules 10, 24, 25: a ceiling claim needs a known-good reference measured on
the same hardware, same data).  The vendor GEMM has NO epilogue (no bias, no GELU, no residual): it is an upper bound for what a
library tile loop reaches at K = 768 / 3072 on this chip, not a replacement.

    python tools/bench_vendor_gemm.py [--pairs 800] [--dtype fp16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402,F401
from rmr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=800)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512
dt = 1 if a.dtype == "fp16" else 0
tdt = torch.float16 if dt else torch.bfloat16
assert lib.rr_set_op_dtype(dt) == 0
g = torch.Generator().manual_seed(0)
for name, N, K in [("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072), ("square", 8192, 8192)]:
    Mx = M if name != "square" else 8192
    A = torch.randn(Mx, K, generator=g).to(tdt).cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).to(tdt).cuda()
    Wt = W.t()                                     # [K, N] view of the K-contiguous weight: the NT form nn.Linear uses
    b = torch.zeros(N).cuda()
    out_v = torch.empty(Mx, N, device="cuda", dtype=tdt)
    out_o = torch.empty(Mx, N, device="cuda", dtype=tdt)

    def ours():
        assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), Mx, N, K, 0, out_o.data_ptr(), st) == 0

    def vendor():
        torch.mm(A, Wt, out=out_v)
    res = {"ours": [], "vendor": []}
    for r in range(a.rounds + 1):
        for nm, fn in (("ours", ours), ("vendor", vendor)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[nm].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * Mx * N * K
    d = (out_o.float() - out_v.float()).abs().max().item()
    print(f"{name:9s} M={Mx} N={N} K={K} {a.dtype}: " + "  ".join(
        f"{nm}: min {min(t):.3f} ms {fl / min(t) / 1e9:7.1f} TF (med {fl / sorted(t)[len(t) // 2] / 1e9:6.1f})" for nm, t in res.items())
        + f"  | max |ours - vendor| {d:.2e}", flush=True)
lib.rr_set_op_dtype(0)
