"""Where the persistent GEMM spends its cycles per output tile, on the PRODUCTION forms of the four big launches of a layer
(folded-LayerNorm QKV / FFN-up + GELU through rr_op_gemm_lnfold, split-residual attention-out / FFN-down through
rr_op_gemm_resid_split), fp16 operands, M = pairs * 512 rows.

    python tools/gemm_epilogue_timeline.py [--pairs 800] [--no-timeline]

Per shape: the launch time of the product kernel (variant 14), then one launch of the diagnostic instantiation (variant 15,
gemm_kernel_hp<..., DIAG = 1>): per-wave s_memtime marks around the sections of the epilogue, summed over the workgroup's
tiles and divided by the tile count.  The diagnostic launch waits for the residual loads in one place (so that "load wait"
and "body" separate); its own total is printed next to the product kernel's so that the distortion is visible.
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
ap.add_argument("--no-timeline", action="store_true")
ap.add_argument("--shapes", default="qkv,attn_out,ffn1,ffn2")
ap.add_argument("--tuning", action="append", default=[], metavar="KEY=INT")
ap.add_argument("--lib", default=None, help="another build of the library (tools/build_variant.py), e.g. tools/bin/librerank_epidiag1.so")
ap.add_argument("--grid", type=int, default=256, help="run the persistent kernels on this many CUs only (rr_set_tuning gemm_grid_cus; a multiple of 8): "
                "what an epilogue costs when fewer CUs share the memory system")
ap.add_argument("--stagger", type=int, default=0, help="rr_set_gemm_stagger: 50..55 = tile-order group of 2..64 row panels, 56 = row-major, 59 = no serpentine K walk")
a = ap.parse_args()
if a.lib:
    import ctypes
    _lib.LIB_PATH = os.path.abspath(a.lib)
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    _lib._SIGS = {k: v for k, v in _lib._SIGS.items() if hasattr(_probe, k)}
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = a.pairs * 512
assert lib.rr_set_op_dtype(1) == 0
assert lib.rr_set_gemm_stagger(a.stagger) == 0
if a.grid != 256:
    assert a.grid % 8 == 0 and lib.rr_set_tuning(b"gemm_grid_cus", a.grid) == 0
for kv in a.tuning:
    k_, v_ = kv.split("=")
    assert lib.rr_set_tuning(k_.encode(), int(v_)) == 0, kv
g = torch.Generator().manual_seed(0)
NAMES = ["main loop", "next-tile setup+prefetch issue", "accumulator arithmetic", "staging writes (sum of passes)",
         "prefetch confirm + touch", "barrier after staging", "residual load issue", "residual load wait", "stream-out body",
         "closing barrier", "tile tail (2 DMA)", "tiles"]
shapes = {"qkv": (2304, 768, "fold", 0), "attn_out": (768, 768, "split", 0), "ffn1": (3072, 768, "fold", 1),
          "ffn2": (768, 3072, "split", 0), "ffn1_plain": (3072, 768, "fold", 0), "qkv_gelu": (2304, 768, "fold", 1),
          "n1536_gelu": (1536, 768, "fold", 1), "n4608": (4608, 768, "fold", 0)}
for name in a.shapes.split(","):
    N, K, kind, epi = shapes[name]
    A = torch.randn(M, K, generator=g).half().cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).half().cuda()
    b = torch.randn(N, generator=g).cuda()
    if kind == "fold":
        csum = W.float().sum(1).contiguous()
        stats = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)

        def run():
            assert lib.rr_op_gemm_lnfold(A.data_ptr(), W.data_ptr(), b.data_ptr(), csum.data_ptr(), stats.data_ptr(), M, N, K, epi,
                                         out.data_ptr(), st) == 0
    else:
        X = (torch.randn(M, N, generator=g) * 3 + 0.5).cuda()
        hi = X.half()
        if "resid_lo8=0" in a.tuning:
            lo = (X - hi.float()).half()
        else:     # the fp16 default: e5m2 bytes of (x - hi) * 16 (timing only: the rows are not in the epilogues' paired-row order)
            lo = torch.zeros((M + 31) // 32 * 32, N, dtype=torch.uint8, device="cuda")
            lo[:M] = ((X - hi.float()) * 16.0).to(torch.float8_e5m2).view(torch.uint8)
        del X
        mu = torch.randn(M, generator=g) * 0.1
        st_in = torch.stack([mu, 1 + 0.1 * torch.rand(M, generator=g)], 1).cuda().contiguous()
        gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
        nparts = (N + 127) // 128
        stats_o, part = torch.empty(M, 2, device="cuda"), torch.empty(M, nparts, 2, device="cuda")

        def run():      # in place, as the forward does
            assert lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), hi.data_ptr(), lo.data_ptr(), st_in.data_ptr(),
                                              gamma.data_ptr(), beta.data_ptr(), M, N, K, 1e-12, hi.data_ptr(), lo.data_ptr(),
                                              stats_o.data_ptr(), part.data_ptr(), st) == 0
    res = {-1: [], 15: []}
    variants = [-1] if a.no_timeline else [-1, 15]          # -1: the shape heuristic = the persistent ring (variant 14)
    for r in range(a.rounds + 1):
        for v in variants:
            assert lib.rr_set_gemm_variant(v) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    line = f"{name:9s} M={M} N={N} K={K}: product {min(res[-1]):.3f} ms {fl / min(res[-1]) / 1e9:7.1f} TF"
    if not a.no_timeline:
        line += f" | diag build {min(res[15]):.3f} ms"
    print(line, flush=True)
    if a.no_timeline:
        continue
    grid = a.grid
    buf = torch.zeros(grid * 8 + grid * 128 + grid * 128 + grid * 64, dtype=torch.int64, device="cuda")
    lib.rr_set_gemm_variant(15)
    lib.rr_set_gemm_stamps(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.rr_set_gemm_stamps(0)
    ep = buf[grid * 8 + grid * 128: grid * 8 + grid * 256].view(grid, 8, 16).double()
    tiles = ep[:, :, 11].clamp(min=1)
    per = ep[:, :, :11] / tiles[:, :, None]            # cycles per tile, [block, wave, section]
    tot = per.sum(-1)
    print(f"   tiles per workgroup {tiles.mean():.2f}; cycles per tile (mean over workgroups and waves): total {tot.mean():.0f}")
    for k in range(11):
        print(f"     {NAMES[k]:34s} {per[:, :, k].mean():8.0f}   by wave " + " ".join(f"{per[:, w, k].mean():7.0f}" for w in range(8)))
    # lockstep check: wall clock (10 ns ticks) of epilogue start / end per workgroup and tile
    tl = buf[grid * 264:].view(grid, 32, 2).double()
    ntl = int(min(tiles.min().item(), 16))
    t0 = tl[:, 0, 0].min()
    print("   epilogue phase across the 256 workgroups (us since the earliest first epilogue): tile: start min/median/max | length median")
    for k in range(ntl):
        st_, en_ = (tl[:, k, 0] - t0) / 100.0, (tl[:, k, 1] - t0) / 100.0
        print(f"     tile {k:2d}: start {st_.min():7.1f} {st_.median():7.1f} {st_.max():7.1f} | epilogue {(en_ - st_).median():6.1f} us"
              + (f" | period {(tl[:, k, 0] - tl[:, k - 1, 0]).median() / 100.0:6.1f} us" if k else ""))
    # fraction of workgroups inside their epilogue, sampled over the steady part of the launch
    lo_, hi_ = tl[:, 2, 0].median(), tl[:, ntl - 2, 0].median()
    grid_t = torch.linspace(lo_.item(), hi_.item(), 400, dtype=torch.float64, device=tl.device)
    inside = ((tl[:, :ntl, 0].reshape(-1, 1) <= grid_t) & (grid_t < tl[:, :ntl, 1].reshape(-1, 1))).view(grid, ntl, -1).any(1).double().mean(0)
    print(f"   workgroups inside their epilogue at a time: mean {inside.mean():.2f}  min {inside.min():.2f}  max {inside.max():.2f}  "
          f"(lockstep: min near 0, max near 1; spread: both near the mean)")
lib.rr_set_gemm_variant(-1)
lib.rr_set_tuning(b"gemm_grid_cus", 0)
lib.rr_set_op_dtype(0)
