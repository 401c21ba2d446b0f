"""GEMM tile-configuration sweep on the GPU (interleaved rounds in ONE process, cdna guide §5.4 rule 24).

    python tools/bench_gemm.py [--pairs 800]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402
from rmr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=800)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="2,10,12,14")
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--stagger", default="0")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--same-rows", default="", help="a / w / aw: every row of A and / or W holds the SAME random values (full memory traffic, "
                "little operand toggling): separates what operand data costs from what operand traffic costs (cf. --stagger 63 / 62)")
ap.add_argument("--timeline", action="store_true", help="variant 13: per-wave phase timeline of the ring kernel (qkv shape)")
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512
shapes = [("qkv", 2304, 768, 0), ("attn_out", 768, 768, 4), ("ffn1", 3072, 768, 1), ("ffn2", 768, 3072, 4)]
variants = [int(v) for v in a.variants.split(",")]
staggers = [int(v) for v in a.stagger.split(",")]
variants = [(v, sg) for v in variants for sg in staggers]
g = torch.Generator().manual_seed(0)
if a.timeline:
    N, K = 2304, 768
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    nwg = (M // 256) * (N // 256)
    buf = torch.zeros(nwg * 8 + nwg * 128, dtype=torch.int64, device="cuda")
    assert lib.rr_set_gemm_variant(13) == 0 and lib.rr_set_gemm_stagger(int(a.stagger.split(',')[0])) == 0
    for _ in range(3):
        lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 0, out.data_ptr(), st)
    lib.rr_set_gemm_stamps(buf.data_ptr())
    assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 0, out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    lib.rr_set_gemm_stamps(0); lib.rr_set_gemm_variant(-1); lib.rr_set_gemm_stagger(0)
    t = buf[: nwg * 8].view(nwg, 8).double()
    print(f"blocks {nwg}: prologue {(t[:,1]-t[:,0]).mean():.0f} main {(t[:,2]-t[:,1]).mean():.0f} epilogue {(t[:,3]-t[:,2]).mean():.0f}")
    d = buf[nwg * 8:].view(nwg, 8, 16).double() / (K // 64)          # per K-tile, [block, wave, segment]
    names = ["p0 reads + BLK a", "p0 read + DMA A1", "p0 BLK b", "p1 read + BLK a", "p1 read + BLK b", "X vmcnt wait",
             "X lgkmcnt(0)", "X s_barrier", "p2 (16 MFMA, 2 DMA)", "p3 (16 MFMA, reads, DMA)", "Y vmcnt wait", "Y lgkmcnt(0)",
             "Y s_barrier"]
    for k, nm in enumerate(names):
        print(f"  {nm:26s} mean {d[:,:,k].mean():7.1f}  by wave " + " ".join(f"{d[:,w,k].mean():6.0f}" for w in range(8)))
    print(f"  sum: {d[:,:,:13].sum(-1).mean():.0f} cycles per K-tile; by wave " + " ".join(f"{d[:,w,:13].sum(-1).mean():6.0f}" for w in range(8)))
    sys.exit(0)
for name, N, K, epi in shapes:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    if "a" in a.same_rows:
        A = A[:1].expand(M, K).contiguous()
    if "w" in a.same_rows:
        W = W[:1].expand(N, K).contiguous()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, device="cuda") if epi == 4 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 4 else torch.bfloat16)

    def run():
        if epi == 4:
            rc = lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, out.data_ptr(), st)
        else:
            rc = lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), st)
        assert rc == 0
    res = {v: [] for v in variants}
    ref = None
    for r in range(a.rounds + 1):
        for v in variants:
            assert lib.rr_set_gemm_variant(v[0]) == 0 and lib.rr_set_gemm_stagger(v[1]) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[v].append(e0.elapsed_time(e1) / 3)
            if r == 0:
                if ref is None:
                    ref = out.float().clone()
                elif not a.no_check:
                    assert (out.float() - ref).abs().max().item() < 1e-2, f"variant {v} disagrees"
    fl = 2.0 * M * N * K
    if a.stamps:
        for v in variants:
            lib.rr_set_gemm_variant(v[0]); lib.rr_set_gemm_stagger(v[1])
            buf = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
            lib.rr_set_gemm_stamps(buf.data_ptr())
            run()
            torch.cuda.synchronize()
            lib.rr_set_gemm_stamps(0)
            t = buf.view(-1, 8)
            t = t[t[:, 3] != 0].double()
            seg = (t[:, 1:4] - t[:, :3])
            print(f"   v{v} stamps over {t.shape[0]} blocks: prologue {seg[:,0].mean():.0f}  main {seg[:,1].mean():.0f}  "
                  f"epilogue {seg[:,2].mean():.0f} cycles (total {(t[:,3]-t[:,0]).mean():.0f}; span all blocks "
                  f"{(t[:,3].max()-t[:,0].min()):.0f}); wave0 sum vmcnt-wait {t[:,4].mean():.0f} barrier-wait {t[:,5].mean():.0f}", flush=True)
    print(f"{name:9s} M={M} N={N} K={K}: " + "  ".join(
        f"v{v}: {min(t):.3f} ms {fl / min(t) / 1e9:7.1f} TF (med {fl / sorted(t)[len(t) // 2] / 1e9:6.1f})" for v, t in res.items()),
        flush=True)
lib.rr_set_gemm_variant(-1); lib.rr_set_gemm_stagger(0)
