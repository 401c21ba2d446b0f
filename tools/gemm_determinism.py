"""Run every bert-base GEMM shape of the bench repeatedly through the production kernel and check the outputs are
bit-identical run to run (a data race in the LDS ring would show up as run-to-run differences)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402
from rmr_amd import _lib  # noqa: E402

lib = _lib.load()
if len(sys.argv) > 2:
    assert lib.rr_set_gemm_variant(int(sys.argv[2])) == 0
st = torch.cuda.current_stream().cuda_stream
M = (int(sys.argv[3]) if len(sys.argv) > 3 else 800) * 512
g = torch.Generator().manual_seed(0)
bad = 0
for name, N, K, epi in [("qkv", 2304, 768, 0), ("attn_out", 768, 768, 4), ("ffn1", 3072, 768, 1), ("ffn2", 768, 3072, 4)]:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, device="cuda") if epi == 4 else None
    ref = None
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 4 else torch.bfloat16)
        if epi == 4:
            rc = lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, out.data_ptr(), st)
        else:
            rc = lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        elif not torch.equal(ref.view(torch.int16 if epi != 4 else torch.int32), out.view(torch.int16 if epi != 4 else torch.int32)):
            ne = (ref != out)
            rows = ne.any(1).nonzero().flatten()
            print(f"{name}: run {it} differs in {int(ne.sum())} elements, rows {rows[:8].tolist()} (tiles {sorted(set((rows // 256).tolist()))[:8]}), "
                  f"max |d| {(ref.float() - out.float()).abs().max().item():.3e}")
            bad += 1
    print(f"{name}: checked", flush=True)
print("NONDETERMINISTIC" if bad else "deterministic")
sys.exit(1 if bad else 0)
