"""Per-wave timeline of the attention kernel (diagnostic instantiation with s_memtime marks): cycles per KV tile spent in
QK^T (LDS reads + 8 MFMAs + next-tile global loads), softmax, PV (transposed LDS reads + 8 MFMAs), staging writes, barrier."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402
from rmr_amd import _lib  # noqa: E402

lib = _lib.load()
if "--online" in sys.argv:                       # A/B: online softmax in every tile instead of the fixed-reference schedule
    sys.argv.remove("--online")
    assert lib.rr_set_tuning(b"attn_fixed_ref", 0) == 0
for arg in list(sys.argv):
    if arg.startswith("--mode="):                # attn_fixed_ref value: 1 = 32 query rows per wave, 2 = 64 (no timeline marks)
        sys.argv.remove(arg)
        assert lib.rr_set_tuning(b"attn_fixed_ref", int(arg[7:])) == 0
st = torch.cuda.current_stream().cuda_stream
B, heads, T = int(sys.argv[1]) if len(sys.argv) > 1 else 800, 12, 512
H = heads * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H, generator=g) * 0.5).bfloat16().cuda()
out = torch.empty(B * T, H, dtype=torch.bfloat16, device="cuda")
kb = torch.zeros(B, T, device="cuda")
nblk = ((B * heads + 7) // 8) * 8 * ((T + 127) // 128)


def run(bias):
    return lib.rr_op_attention_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, 3 * H,
                                    bias, B, heads, T, T, 1, out.data_ptr(), H, st)


for name, bias in (("no mask", 0), ("masked path", kb.data_ptr()), ("no valid key: every workgroup recomputed online", kb.data_ptr())):
    if bias:
        kb[:, 400:] = -1e30
    if name.startswith("no valid"):
        kb[:] = -1e30
    for _ in range(2):
        assert run(bias) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(bias)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = torch.zeros(nblk * 32, dtype=torch.int64, device="cuda")
    lib.rr_set_attn_stamps(buf.data_ptr())
    assert run(bias) == 0
    torch.cuda.synchronize()
    lib.rr_set_attn_stamps(0)
    d = buf.view(nblk, 4, 8).double()
    nt = d[:, 0, 7].clamp(min=1)[:, None, None]
    seg = (d[:, :, :5] / nt)
    names = ["QK^T (+ next-tile loads)", "softmax", "PV", "staging writes", "barrier"]
    print(f"{name}: {ms:.3f} ms, {4.0 * B * heads * T * T * 64 / ms / 1e9:.0f} TFLOP/s")
    for k, nm in enumerate(names):
        print(f"  {nm:26s} {seg[:, :, k].mean():7.1f} cycles per KV tile   by wave " + " ".join(f"{seg[:, w, k].mean():6.0f}" for w in range(4)))
    print(f"  sum {seg.sum(-1).mean():.0f} cycles per KV tile per wave")
