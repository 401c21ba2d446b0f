"""Per-shape rate of the e4m3 GEMM (rr_op_gemm_fp8) next to the 16-bit production GEMM on the same shapes.

    python tools/bench_gemm_fp8.py [--pairs 800]
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
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512
# (name, N, K, fp8 epilogue, bf16 epilogue)
shapes = [("qkv 768", 2304, 768, 0, 0), ("ffn1 768", 3072, 768, 1, 1), ("qkv 1024 (bert-large)", 3072, 1024, 0, 0),
          ("ffn1 1024 (bert-large)", 4096, 1024, 1, 1), ("ffn2 1024 (bert-large, bf16 out)", 1024, 4096, 0, 0)]
g = torch.Generator().manual_seed(0)
for name, N, K, e8, e16 in shapes:
    a8 = torch.randint(0, 256, (M, K), dtype=torch.uint8, generator=g)
    a8[(a8 & 0x7f) == 0x7f] = 0x38                     # no NaN encodings
    w8 = torch.randint(0, 256, (N, K), dtype=torch.uint8, generator=g)
    w8[(w8 & 0x7f) == 0x7f] = 0x38
    a8, w8 = a8.cuda(), w8.cuda()
    a16 = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w16 = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {}
    for _ in range(a.rounds):
        for kind in ("fp8", "bf16"):
            def run():
                if kind == "fp8":
                    return lib.rr_op_gemm_fp8(a8.data_ptr(), w8.data_ptr(), bias.data_ptr(), 1e-4, M, N, K, e8, out.data_ptr(), st)
                return lib.rr_op_gemm_bf16(a16.data_ptr(), w16.data_ptr(), bias.data_ptr(), M, N, K, e16, out.data_ptr(), st)
            assert run() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(kind, []).append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * M * N * K
    print(f"{name:34s} M={M} N={N} K={K}: " + "  ".join(f"{k}: {min(v):.3f} ms {fl / min(v) / 1e9:7.1f} TF" for k, v in res.items()))
