"""Persistent-ring GEMM, staged epilogue (variant 14, gemm_kernel_hp) against the direct epilogue (variant 16, gemm_kernel_hq)
on the production 16-bit forms (folded-LayerNorm QKV / FFN-up + GELU through rr_op_gemm_lnfold, fp16 operands), interleaved
rounds in ONE process (cdna guide §5.4 rule 24), outputs compared bit for bit; then one launch of the diagnostic build of hq
(variant 17): s_memtime sections per tile.

    python tools/gemm_direct_ab.py [--pairs 800] [--shapes qkv,ffn1] [--dtype fp16]
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
ap.add_argument("--shapes", default="qkv,ffn1")
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--no-timeline", action="store_true")
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512
dt = 1 if a.dtype == "fp16" else 0
assert lib.rr_set_op_dtype(dt) == 0
cast = (lambda t: t.half()) if dt else (lambda t: t.bfloat16())
g = torch.Generator().manual_seed(0)
shapes = {"qkv": (2304, 768, 0), "ffn1": (3072, 768, 1), "kv_ce": (1536, 768, 0), "qkv_large": (3072, 1024, 0),
          "ffn1_large": (4096, 1024, 1)}
NAMES = ["main loop", "next-tile prefetch issue (16 DMA + params)", "accumulator arithmetic", "pack + 16 stores",
         "tile start: wait K-tile 0 + barrier"]
for name in a.shapes.split(","):
    N, K, epi = shapes[name]
    A = cast(torch.randn(M, K, generator=g)).cuda()
    W = cast(torch.randn(N, K, generator=g) * 0.02).cuda()
    b = torch.randn(N, generator=g).cuda()
    csum = W.float().sum(1).contiguous()
    stats = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
    outs = {v: torch.empty(M, N, device="cuda", dtype=A.dtype) for v in (14, 16)}

    def run(v):
        assert lib.rr_op_gemm_lnfold(A.data_ptr(), W.data_ptr(), b.data_ptr(), csum.data_ptr(), stats.data_ptr(), M, N, K, epi,
                                     outs[v if v in outs else 16].data_ptr(), st) == 0
    res = {14: [], 16: []}
    for r in range(a.rounds + 1):
        for v in (14, 16):
            assert lib.rr_set_gemm_variant(v) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[v].append(e0.elapsed_time(e1) / 3)
    same = torch.equal(outs[14].view(torch.int16), outs[16].view(torch.int16))
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M} N={N} K={K} {a.dtype}: " + "  ".join(
        f"v{v}: min {min(t):.3f} ms {fl / min(t) / 1e9:7.1f} TF (med {sorted(t)[len(t) // 2]:.3f})" for v, t in res.items())
        + f"  | bitwise {'EQUAL' if same else 'DIFFERENT'}", flush=True)
    if a.no_timeline:
        continue
    grid = 256
    buf = torch.zeros(grid * 8 * 8, dtype=torch.int64, device="cuda")
    assert lib.rr_set_gemm_variant(17) == 0
    lib.rr_set_gemm_stamps(buf.data_ptr())
    run(17)
    torch.cuda.synchronize()
    lib.rr_set_gemm_stamps(0)
    ep = buf.view(grid, 8, 8).double()
    tiles = ep[:, :, 5].clamp(min=1)
    per = ep[:, :, :5] / tiles[:, :, None]
    print(f"   hq diagnostic build: tiles per workgroup {tiles.mean():.2f}; cycles per tile {per.sum(-1).mean():.0f}")
    for k in range(5):
        print(f"     {NAMES[k]:44s} {per[:, :, k].mean():8.0f}   by wave " + " ".join(f"{per[:, w, k].mean():7.0f}" for w in range(8)))
lib.rr_set_gemm_variant(-1)
lib.rr_set_op_dtype(0)
