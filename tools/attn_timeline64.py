"""Per-wave timeline of the 64-rows-per-wave fixed-reference attention kernel (attn_fixed64_kernel<0, DIAG64>): s_memtime ticks
per KV tile in each section, summed over a workgroup's tiles (bf16 operands, 800 x 12 heads, T = 512 by default).

    python tools/attn_timeline64.py [B] [--tuning KEY=INT ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402,F401
from rmr_amd import _lib  # noqa: E402

lib = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
for i, a in enumerate(sys.argv):
    if a == "--tuning":
        k, v = sys.argv[i + 1].split("=")
        assert lib.rr_set_tuning(k.encode(), int(v)) == 0
args = [a for a in args if "=" not in a]
assert lib.rr_set_tuning(b"attn_fixed_ref", 2) == 0
st = torch.cuda.current_stream().cuda_stream
B, heads, T = int(args[0]) if args else 800, 12, 512
H = heads * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H, generator=g) * 0.5).bfloat16().cuda()
out = torch.empty(B * T, H, dtype=torch.bfloat16, device="cuda")
nblk = ((B * heads + 7) // 8) * 8 * ((T + 255) // 256)
kb = torch.zeros(B, T, device="cuda") if "--bias" in sys.argv else None       # an all-valid key bias, as the forward passes


def run():
    return lib.rr_op_attention_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, 3 * H, kb.data_ptr() if kb is not None else None, B, heads, T, T, 1,
                                    out.data_ptr(), H, st)


for _ in range(30):                 # the first ~10 launches of a process run at a lower clock: 1.15 ms against 0.91 warm
    assert run() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
buf = torch.zeros(nblk * 32, dtype=torch.int64, device="cuda")
lib.rr_set_attn_stamps(buf.data_ptr())
assert run() == 0
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms_d = e0.elapsed_time(e1) / 5
lib.rr_set_attn_stamps(0)
for _ in range(3):
    run()
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"product again, after the diagnostic launches (20 launches): {e0.elapsed_time(e1) / 20:.3f} ms")
d = buf.view(nblk, 4, 8).double()
d = d[d[:, 0, 6] > 0]
nt = d[:, :, 6:7]
seg = d[:, :, [0, 1, 2, 3, 5]] / nt
names = ["next-tile DMA issue + K fragment requests", "QK^T issue (+ V fragment requests)", "softmax (waits for QK^T)", "pack + P.V issue",
         "next tile's DMA wait + workgroup barrier"]
print(f"product {ms:.3f} ms ({4.0 * B * heads * T * T * 64 / ms / 1e9:.0f} TFLOP/s) | diagnostic build {ms_d:.3f} ms; {d.shape[0]} workgroups, {nt.mean():.1f} tiles each")
for k, nm in enumerate(names):
    print(f"  {nm:44s} {seg[:, :, k].mean():7.1f} ticks per KV tile   by wave " + " ".join(f"{seg[:, w, k].mean():6.0f}" for w in range(4)))
loop = (seg.sum(-1) * nt[:, :, 0]).mean()
print(f"  tile loop {loop:.0f} ticks per block ({seg.sum(-1).mean():.0f} per tile); prologue {d[:, :, 4].mean():.0f}; output (normalise, stage, store) "
      f"{(d[:, :, 7] - d[:, :, 4]).mean() - loop:.0f}; whole block {d[:, :, 7].mean():.0f}")
