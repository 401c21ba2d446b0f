"""One BERT layer out of the stand-alone operator entry points, repeated: which operator is not reproducible?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402
from rmr_amd import _lib  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, S, H, I, heads = 200, 512, 768, 3072, 12
M = n * S
g = torch.Generator().manual_seed(0)
x32 = torch.randn(M, H, generator=g).cuda()
wqkv = (torch.randn(3 * H, H, generator=g) * 0.02).bfloat16().cuda()
wo = (torch.randn(H, H, generator=g) * 0.02).bfloat16().cuda()
w1 = (torch.randn(I, H, generator=g) * 0.02).bfloat16().cuda()
w2 = (torch.randn(H, I, generator=g) * 0.02).bfloat16().cuda()
bq, bo, b1, b2 = (torch.randn(k, generator=g).cuda() * 0.1 for k in (3 * H, H, I, H))
gam, bet = torch.ones(H).cuda(), torch.zeros(H).cuda()
lens = torch.randint(64, S + 1, (n,), generator=g)
kb = torch.zeros(n, S)
for i in range(n):
    kb[i, lens[i]:] = -1e30
kb = kb.cuda()


def layer():
    out = {}
    h16 = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    h32 = torch.empty(M, H, device="cuda")
    assert lib.rr_op_layernorm(x32.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-12, M, H, h32.data_ptr(), h16.data_ptr(), st) == 0
    out["ln0"] = h16
    qkv = torch.empty(M, 3 * H, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_gemm_bf16(h16.data_ptr(), wqkv.data_ptr(), bq.data_ptr(), M, 3 * H, H, 0, qkv.data_ptr(), st) == 0
    out["qkv"] = qkv
    ctx = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_attention_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, 3 * H, kb.data_ptr(), n,
                                    heads, S, S, 1, ctx.data_ptr(), H, st) == 0
    out["attn"] = ctx
    pre = torch.empty(M, H, device="cuda")
    assert lib.rr_op_gemm_resid_f32(ctx.data_ptr(), wo.data_ptr(), bo.data_ptr(), h32.data_ptr(), M, H, H, pre.data_ptr(), st) == 0
    out["attn_out"] = pre
    a16 = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    a32 = torch.empty(M, H, device="cuda")
    assert lib.rr_op_layernorm(pre.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-12, M, H, a32.data_ptr(), a16.data_ptr(), st) == 0
    out["ln1"] = a16
    mid = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_gemm_bf16(a16.data_ptr(), w1.data_ptr(), b1.data_ptr(), M, I, H, 1, mid.data_ptr(), st) == 0
    out["ffn1"] = mid
    pre2 = torch.empty(M, H, device="cuda")
    assert lib.rr_op_gemm_resid_f32(mid.data_ptr(), w2.data_ptr(), b2.data_ptr(), a32.data_ptr(), M, H, I, pre2.data_ptr(), st) == 0
    out["ffn2"] = pre2
    # the same block in the LayerNorm-statistics dataflow
    stats = torch.empty(M, 2, device="cuda")
    b16 = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_layernorm_stats(pre.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-12, M, H, 0, b16.data_ptr(), stats.data_ptr(), st) == 0
    out["ln1_stats"] = stats
    out["ln1_b16"] = b16
    pre3 = torch.empty(M, H, device="cuda")
    assert lib.rr_op_gemm_ln_resid_f32(mid.data_ptr(), w2.data_ptr(), b2.data_ptr(), pre.data_ptr(), stats.data_ptr(), gam.data_ptr(),
                                       bet.data_ptr(), M, H, I, pre3.data_ptr(), st) == 0
    out["ffn2_ln"] = pre3
    torch.cuda.synchronize()
    return out


lib.rr_set_gemm_variant(variant)
if len(sys.argv) > 2:
    assert lib.rr_set_gemm_stagger(int(sys.argv[2])) == 0
ref = layer()
for it in range(8):
    o = layer()
    msg = []
    for k in ref:
        a, b = ref[k].float(), o[k].float()
        ne = (a != b)
        if ne.any():
            rows = ne.any(1).nonzero().flatten()
            msg.append(f"{k}: {int(ne.sum())} elems / {len(rows)} rows differ (first row {int(rows[0])}, max {(a - b).abs().max().item():.2e})")
    print(f"variant {variant} run {it}: " + ("; ".join(msg) if msg else "identical"), flush=True)
lib.rr_set_gemm_variant(-1)

# detail of the LN-residual GEMM under the chosen variant: where do runs differ, and which run is right?
lib.rr_set_gemm_variant(variant)
o1, o2 = layer(), layer()
for tries in range(6):
    if not torch.equal(o1["ffn2_ln"], o2["ffn2_ln"]):
        break
    o2 = layer()
a, b = o1["ffn2_ln"], o2["ffn2_ln"]
ne = (a != b).nonzero()
if len(ne):
    rows, cols = ne[:, 0], ne[:, 1]
    print("differing rows", rows.min().item(), "..", rows.max().item(), "tiles(128)", sorted(set((rows // 128).tolist())),
          "cols", sorted(set(cols.tolist()))[:16])
    st_ = o1["ln1_stats"]
    x = o1["attn_out"]
    expect = o1["ffn2"]          # same GEMM with the materialised fp32 residual
    for r, c in ne[:6].tolist():
        print(f"  ({r},{c}): run A {a[r, c].item():+.5f}  run B {b[r, c].item():+.5f}  plain-residual path {expect[r, c].item():+.5f}  "
              f"x {x[r, c].item():+.4f} mean {st_[r, 0].item():+.4f} rstd {st_[r, 1].item():.4f}")
lib.rr_set_gemm_variant(-1)
