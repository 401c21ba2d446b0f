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
    # wave lifetimes and placement: clock = cycles between the first entry and the last exit / the launch's wall time;
    # residency = sum of lifetimes / (span x SIMDs that hosted a wave)
    raw = buf.view(nblk, 4, 8)
    live = raw[:, 1:, 5] > 0
    t0, t1, where = raw[:, 1:, 5][live], raw[:, 1:, 6][live], raw[:, 1:, 7][live]
    if t0.numel():
        lib.rr_set_attn_stamps(buf.data_ptr())
        run(bias); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            run(bias)
        ev1.record(); torch.cuda.synchronize()
        lib.rr_set_attn_stamps(0)
        ms_d = ev0.elapsed_time(ev1) / 5
        xcc = (where >> 32) & 0xf
        hw = where & 0xffffffff
        simd_key = (xcc << 16) | (hw & 0xff30)               # HW_ID: SIMD [5:4], CU [11:8], SH [12], SE [15:13]
        keys = torch.unique(simd_key)
        life = (t1 - t0).double()
        spans, resident = [], []
        for kx in keys[:: max(1, keys.numel() // 64)].tolist():      # a sample of SIMDs: one time base each
            m = simd_key == kx
            sp = float((t1[m].max() - t0[m].min()).item())
            spans.append(sp)
            resident.append(life[m].sum().item() * 4 / 3 / sp)
        spans, resident = torch.tensor(spans), torch.tensor(resident)
        print(f"  diagnostic build, 5 launches back to back: {ms_d:.3f} ms each; per-SIMD span median {spans.median():.0f} cycles "
              f"(min {spans.min():.0f}, max {spans.max():.0f}) -> {spans.median().item() / ms_d / 1e6:.2f} GHz if the span is the launch")
        print(f"  wave lifetime mean {life.mean():.0f} cycles (tile loop {seg.sum(-1).mean() * float(nt.mean()):.0f}); {keys.numel()} distinct (XCC, SE, SH, CU, SIMD) ids; "
              f"resident waves per SIMD: median {resident.median():.2f} (min {resident.min():.2f}, max {resident.max():.2f})")
