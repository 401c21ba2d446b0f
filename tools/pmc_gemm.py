"""One GEMM shape, one variant, a few launches: target for `rocprofv3 --pmc ...` (python3 tools/pmc_gemm.py V)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd
from rmr_amd import _lib
lib = _lib.load()
v = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, K, epi = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2304, 768, 0)
M = 409600
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).bfloat16().cuda()
W = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
b = torch.randn(N, generator=g).cuda()
out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
lib.rr_set_gemm_variant(v)
if os.environ.get("RR_STAGGER"):            # tile-order / K-walk code of the persistent ring (rr_set_gemm_stagger)
    assert lib.rr_set_gemm_stagger(int(os.environ["RR_STAGGER"])) == 0
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), st) == 0
torch.cuda.synchronize()
