"""Time the product attention launch (rr_op_attention_bf16: the 64-row fixed-reference form + its normally empty redo launch) at the
bench shape, for ONE build of the library — the child of an A/B between builds (alternate the libraries in a shell loop).

    python tools/bench_attention.py [LIB.so] [--pairs 800] [--heads 12] [--T 512] [--launches 40]

Prints per operand type the min / median launch time over `--rounds` groups of `--launches` and a hash of the output rows.
"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("lib", nargs="?")
ap.add_argument("--pairs", type=int, default=800)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
from rmr_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
    import ctypes
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    _lib._SIGS = {k: v for k, v in _lib._SIGS.items() if hasattr(_probe, k)}
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B, heads, T = a.pairs, a.heads, a.T
H = heads * 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B * T, 3 * H, generator=g) * 0.5
kb = torch.zeros(B, T, device="cuda")            # an all-valid key bias, as the forward passes
for dt, tdt in ((1, torch.float16), (0, torch.bfloat16)):
    assert lib.rr_set_op_dtype(dt) == 0
    qkv = x.to(tdt).cuda()
    out = torch.empty(B * T, H, dtype=tdt, device="cuda")

    def run():
        assert lib.rr_op_attention_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, 3 * H, kb.data_ptr(), B, heads, T, T, 1,
                                        out.data_ptr(), H, st) == 0
    for _ in range(30):                              # the first launches of a process run at a lower clock
        run()
    ts = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.launches)
    ts.sort()
    fl = 4.0 * B * heads * T * T * 64
    print(f"{os.path.basename(_lib.LIB_PATH):28s} {'fp16' if dt else 'bf16'}: min {ts[0]:.4f} ms  median {ts[len(ts) // 2]:.4f} ms  ({fl / ts[0] / 1e9:.0f} TFLOP/s)  "
          f"out sha {hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
lib.rr_set_op_dtype(0)
