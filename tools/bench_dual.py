"""The dual-workgroup GEMM (variant 16: two 4-wave workgroups per CU on 256x128 tiles) against the persistent ring (variant 14) on the
production forms of a layer's four big launches: bitwise comparison of the outputs, then interleaved timing in one process.

    python tools/bench_dual.py [--pairs 800] [--skews -1,0,8,24] [--dtype fp16]
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
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--skews", default="-1")
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--shapes", default="qkv,attn_out,ffn1,ffn2")
ap.add_argument("--ragged", type=int, default=0, help="rows removed from M (ragged last row tile)")
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512 - a.ragged
dt = 1 if a.dtype == "fp16" else 0
t16 = torch.float16 if dt else torch.bfloat16
assert lib.rr_set_op_dtype(dt) == 0
g = torch.Generator().manual_seed(0)
shapes = {"qkv": (2304, 768, "fold", 0), "attn_out": (768, 768, "split", 0), "ffn1": (3072, 768, "fold", 1),
          "ffn2": (768, 3072, "split", 0), "plain": (2304, 768, "plain", 0)}
skews = [int(v) for v in a.skews.split(",")]
ok_all = True
for name in a.shapes.split(","):
    N, K, kind, epi = shapes[name]
    A = torch.randn(M, K, generator=g).to(t16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).to(t16).cuda()
    b = torch.randn(N, generator=g).cuda()
    if kind in ("fold", "plain"):
        csum = W.float().sum(1).contiguous()
        stats = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
        out = torch.empty(M, N, device="cuda", dtype=t16)
        outs = [out]

        def run():
            if kind == "fold":
                rc = lib.rr_op_gemm_lnfold(A.data_ptr(), W.data_ptr(), b.data_ptr(), csum.data_ptr(), stats.data_ptr(), M, N, K, epi,
                                           out.data_ptr(), st)
            else:
                rc = lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), st)
            assert rc == 0, rc
    else:
        X = (torch.randn(M, N, generator=g) * 3 + 0.5).cuda()
        hi = X.to(t16)
        lo = (X - hi.float()).half()
        del X
        st_in = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
        gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
        nparts = (N + 127) // 128
        x16, lo_out = torch.empty_like(hi), torch.empty_like(lo)
        stats_o, part = torch.empty(M, 2, device="cuda"), torch.empty(M, nparts, 2, device="cuda")
        outs = [x16, lo_out, stats_o]

        def run():      # out of place here, so that every run sees the same residual rows
            rc = lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), hi.data_ptr(), lo.data_ptr(), st_in.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), M, N, K, 1e-12, x16.data_ptr(), lo_out.data_ptr(),
                                            stats_o.data_ptr(), part.data_ptr(), st)
            assert rc == 0, rc
    # ---- bitwise comparison
    for o in outs:
        o.zero_()
    assert lib.rr_set_gemm_variant(-1) == 0
    run()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    for o in outs:
        o.fill_(7)
    assert lib.rr_set_gemm_variant(16) == 0
    run()
    torch.cuda.synchronize()
    same = all(torch.equal(o.view(torch.int16) if o.dtype != torch.float32 else o.view(torch.int32),
                           r.view(torch.int16) if r.dtype != torch.float32 else r.view(torch.int32)) for o, r in zip(outs, ref))
    if not same:
        ok_all = False
        for o, r in zip(outs, ref):
            d = (o.float() - r.float()).abs()
            bad = (d > 0).nonzero()
            print(f"   MISMATCH {name}: max |d| {d.max().item():.3e}, {bad.shape[0]} elements differ; first {bad[:5].tolist()}")
    # ---- timing
    cfgs = [(-1, 0)] + [(16, s) for s in skews]
    res = {c: [] for c in cfgs}
    for r in range(a.rounds + 1):
        for c in cfgs:
            assert lib.rr_set_gemm_variant(c[0]) == 0
            if c[0] == 16:
                assert lib.rr_set_tuning(b"gemm_dual_skew", c[1]) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[c].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K} bitwise {'OK' if same else 'DIFF'}: " + "  ".join(
        f"{'ring' if c[0] < 0 else 'dual skew ' + str(c[1])}: {min(t):.3f} ms {fl / min(t) / 1e9:6.1f} TF" for c, t in res.items()), flush=True)
lib.rr_set_gemm_variant(-1)
lib.rr_set_tuning(b"gemm_dual_skew", -1)
lib.rr_set_op_dtype(0)
sys.exit(0 if ok_all else 1)
