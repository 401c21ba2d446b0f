"""One attention shape, a few launches: target for `rocprofv3 --pmc ...`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd
from rmr_amd import _lib
lib = _lib.load()
B, heads, T = 800, 12, 512
H = heads * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H, generator=g) * 0.5).bfloat16().cuda()
out = torch.empty(B, T, H, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
base = qkv.data_ptr()
for _ in range(3):
    assert lib.rr_op_attention_bf16(base, base + 2 * H, base + 4 * H, 3 * H, 3 * H, 0, B, heads, T, T, 1, out.data_ptr(), H, st) == 0
torch.cuda.synchronize()
