"""Persistent-ring GEMM with the weight operand through LDS (gemm_wreg 0) against straight from L2 into registers (gemm_wreg 1,
gemm_kernel_hp<..., WREG>) on the PRODUCTION forms (folded-LayerNorm QKV / FFN-up + GELU through rr_op_gemm_lnfold, split-residual
attention-out / FFN-down through rr_op_gemm_resid_split), interleaved rounds in one process, outputs compared bit for bit.

    python tools/gemm_wreg_ab.py [--pairs 800] [--dtype fp16]
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
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--shapes", default="qkv,attn_out,ffn1,ffn2")
ap.add_argument("--stagger", type=int, default=0, help="60 / 61: diagnostic builds of the WREG loop (wrong results)")
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512
dt = 1 if a.dtype == "fp16" else 0
assert lib.rr_set_op_dtype(dt) == 0
assert lib.rr_set_gemm_stagger(a.stagger) == 0
cast = (lambda t: t.half()) if dt else (lambda t: t.bfloat16())
g = torch.Generator().manual_seed(0)
shapes = {"qkv": (2304, 768, "fold", 0), "attn_out": (768, 768, "split", 0), "ffn1": (3072, 768, "fold", 1), "ffn2": (768, 3072, "split", 0)}
for name in a.shapes.split(","):
    N, K, kind, epi = shapes[name]
    A = cast(torch.randn(M, K, generator=g)).cuda()
    W = cast(torch.randn(N, K, generator=g) * 0.02).cuda()
    b = torch.randn(N, generator=g).cuda()
    if kind == "fold":
        csum = W.float().sum(1).contiguous()
        stats = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
        outs = {v: torch.empty(M, N, device="cuda", dtype=A.dtype) for v in (0, 1)}

        def run(v):
            assert lib.rr_op_gemm_lnfold(A.data_ptr(), W.data_ptr(), b.data_ptr(), csum.data_ptr(), stats.data_ptr(), M, N, K, epi,
                                         outs[v].data_ptr(), st) == 0

        def same():
            return torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    else:
        X = (torch.randn(M, N, generator=g) * 3 + 0.5).cuda()
        hi0 = cast(X)
        lo0 = (X - hi0.float()).half()
        del X
        st_in = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
        gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
        nparts = (N + 127) // 128
        o = {v: (torch.empty_like(hi0), torch.empty_like(lo0), torch.empty(M, 2, device="cuda"), torch.empty(M, nparts, 2, device="cuda")) for v in (0, 1)}

        def run(v):      # out of place, so that both settings see the same residual rows
            assert lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), hi0.data_ptr(), lo0.data_ptr(), st_in.data_ptr(),
                                              gamma.data_ptr(), beta.data_ptr(), M, N, K, 1e-12, o[v][0].data_ptr(), o[v][1].data_ptr(),
                                              o[v][2].data_ptr(), o[v][3].data_ptr(), st) == 0

        def same():
            return all(torch.equal(x.view(torch.int16) if x.dtype != torch.float32 else x, y.view(torch.int16) if y.dtype != torch.float32 else y)
                       for x, y in zip(o[0][:3], o[1][:3]))
    res = {0: [], 1: []}
    for r in range(a.rounds + 1):
        for v in (0, 1):
            assert lib.rr_set_tuning(b"gemm_wreg", v) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K} {a.dtype}: " + "  ".join(
        f"wreg={v}: min {min(t):.3f} ms {fl / min(t) / 1e9:7.1f} TF (med {sorted(t)[len(t) // 2]:.3f})" for v, t in res.items())
        + f"  | bitwise {'EQUAL' if same() else 'DIFFERENT'}", flush=True)
lib.rr_set_tuning(b"gemm_wreg", 0)
lib.rr_set_gemm_stagger(0)
lib.rr_set_op_dtype(0)
