"""GPU: the e4m3 GEMM on the block-scaled matrix core (rr_op_gemm_fp8, csrc/gemm_fp8.hip) against a plain PyTorch fp32
reference of the same quantised operands.  Products of two e4m3 values are exact in fp32, so the only difference is the
accumulation inside the matrix core (128 products per instruction are not summed at full fp32 precision: measured up to
2.5e-5 of the absolute dot product) and its order: tolerance 1e-4 of the row's absolute dot product (+ bf16 output
rounding where it applies) — three orders of magnitude below the e4m3 quantisation step itself."""
import numpy as np
import pytest
import torch

from helpers import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from rmr_amd import _lib
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _quant(x):
    s = x.abs().max().item() / 448.0                       # e4m3fn max normal
    q = (x / s).to(torch.float8_e4m3fn)
    return q, s


@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 128, 2), (300, 768, 768, 0), (1000, 3072, 1024, 1), (37, 260, 256, 2),
                                       (513, 1024, 4096, 0), (4096, 2304, 768, 2)])
def test_gemm_fp8_matches_fp32_reference(lib, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 1.7
    w = torch.randn(N, K, generator=g) * 0.04
    bias = torch.randn(N, generator=g) * 0.1
    a8, sa = _quant(a)
    w8, sw = _quant(w)
    af, wf = a8.float(), w8.float()
    ref = (af @ wf.T) * (sa * sw) + bias
    mag = (af.abs() @ wf.abs().T) * (sa * sw) + bias.abs()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    out = torch.full((M, N), float("nan"), dtype=torch.float32 if epi == 2 else torch.bfloat16, device="cuda")
    a_d, w_d, b_d = a8.view(torch.uint8).cuda(), w8.view(torch.uint8).cuda(), bias.cuda()
    rc = lib.rr_op_gemm_fp8(a_d.data_ptr(), w_d.data_ptr(), b_d.data_ptr(), sa * sw, M, N, K, epi, out.data_ptr(), _stream())
    assert rc == 0
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    tol = 1e-4 * mag + (0 if epi == 2 else 2.0 ** -8 * ref.abs()) + 1e-6
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} beyond tolerance, max |d| {(got - ref).abs().max().item():.3e}"


def test_gemm_fp8_rejects_bad_shapes(lib):
    x = torch.zeros(256, 256, dtype=torch.uint8, device="cuda")
    out = torch.zeros(256, 256, device="cuda")
    assert lib.rr_op_gemm_fp8(x.data_ptr(), x.data_ptr(), 0, 1.0, 256, 256, 192, 2, out.data_ptr(), _stream()) != 0   # K % 128
    assert lib.rr_op_gemm_fp8(x.data_ptr(), x.data_ptr(), 0, 1.0, 256, 254, 128, 2, out.data_ptr(), _stream()) != 0   # N % 4
    assert lib.rr_op_gemm_fp8(x.data_ptr(), x.data_ptr(), 0, 1.0, 256, 256, 128, 7, out.data_ptr(), _stream()) != 0   # epilogue


@pytest.mark.parametrize("src", ["f32", "bf16"])
def test_quantize_and_amax_match_torch(lib, src):
    g = torch.Generator().manual_seed(5)
    n = 8 * 12345
    x = torch.randn(n, generator=g) * 3.0
    x[::97] *= 40.0                                             # values beyond the e4m3 range after scaling: saturate
    x[5] = 0.0
    xd = x.cuda() if src == "f32" else x.bfloat16().cuda()
    xs = xd.float().cpu()                                       # what the device actually sees
    amax = torch.zeros(1, device="cuda")
    assert lib.rr_op_amax(xd.data_ptr(), int(src == "f32"), n, amax.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert amax.item() == xs.abs().max().item()
    scale = 0.05                                                # deliberately too small: |x| / scale exceeds 448 for some
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert lib.rr_op_quantize_fp8(xd.data_ptr(), int(src == "f32"), scale, out.data_ptr(), n, _stream()) == 0
    torch.cuda.synchronize()
    want = (xs * (1.0 / scale)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)   # same 1/scale multiply as the device
    got = out.cpu()
    diff = got != want
    assert not diff.any(), f"{int(diff.sum())} of {n} bytes differ, first at {int(diff.nonzero()[0])}"


def test_device_quantised_gemm_end_to_end(lib):
    """amax -> scale -> quantise on the device -> e4m3 GEMM, against the fp32 product of the unquantised operands:
    the error is the e4m3 quantisation noise (2^-4 relative per element, averaging over K)."""
    M, N, K = 512, 768, 1024
    g = torch.Generator().manual_seed(6)
    a = (torch.randn(M, K, generator=g) * 0.8).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).cuda()
    amax = torch.zeros(2, device="cuda")
    assert lib.rr_op_amax(a.data_ptr(), 0, a.numel(), amax.data_ptr(), _stream()) == 0
    assert lib.rr_op_amax(w.data_ptr(), 1, w.numel(), amax.data_ptr() + 4, _stream()) == 0
    sa, sw = (amax.cpu() / 448.0).tolist()
    a8 = torch.empty(M, K, dtype=torch.uint8, device="cuda")
    w8 = torch.empty(N, K, dtype=torch.uint8, device="cuda")
    assert lib.rr_op_quantize_fp8(a.data_ptr(), 0, sa, a8.data_ptr(), a.numel(), _stream()) == 0
    assert lib.rr_op_quantize_fp8(w.data_ptr(), 1, sw, w8.data_ptr(), w.numel(), _stream()) == 0
    out = torch.empty(M, N, device="cuda")
    assert lib.rr_op_gemm_fp8(a8.data_ptr(), w8.data_ptr(), 0, sa * sw, M, N, K, 2, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    ref = a.float() @ w.T
    rel = ((out - ref).norm() / ref.norm()).item()
    print(f"[fp8 gemm, device-quantised] relative Frobenius error vs the 16-bit-operand product: {rel:.3e}")
    assert rel < 0.05


@pytest.mark.parametrize("M,N,K,epi", [(33_000, 1024, 128, 0), (33_000, 1024, 256, 1), (33_100, 1024, 1024, 0),
                                       (66_000, 768, 768 + 128, 1), (300, 384, 256, 2), (517, 1024, 384, 0)])
def test_gemm_fp8_row_and_channel_scales(lib, M, N, K, epi):
    """rr_op_gemm_fp8_rc: per-row activation scales x per-channel weight scales.  M >= 33 000 at N = 1024 is 516 tiles of
    256x256: the persistent ring on the 32x32x64 MFMA, with one, two and many K-tiles (prologue / tail / steady-state
    paths) and a partial last row tile; the small shapes run the two-stage kernel."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g) * (0.2 + 3.0 * torch.rand(M, 1, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g) * (0.01 + 0.08 * torch.rand(N, 1, device="cuda", generator=g))
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    sa = a.abs().amax(1) / 448.0
    sw = w.abs().amax(1) / 448.0
    a8 = (a / sa[:, None]).to(torch.float8_e4m3fn)
    w8 = (w / sw[:, None]).to(torch.float8_e4m3fn)
    out = torch.full((M, N), float("nan"), dtype=torch.float32 if epi == 2 else torch.bfloat16, device="cuda")
    rc = lib.rr_op_gemm_fp8_rc(a8.view(torch.uint8).data_ptr(), w8.view(torch.uint8).data_ptr(), bias.data_ptr(),
                               sa.data_ptr(), sw.data_ptr(), M, N, K, epi, out.data_ptr(), _stream())
    assert rc == 0
    torch.cuda.synchronize()
    for r0 in sorted({0, max(0, M // 2 - 150), max(0, M - 300)}):        # row blocks at the start, the middle and the partial last tile
        rows = slice(r0, min(M, r0 + 300))
        af, wf = a8[rows].float(), w8.float()
        ref = (af @ wf.T) * sa[rows, None] * sw[None, :] + bias
        mag = (af.abs() @ wf.abs().T) * sa[rows, None] * sw[None, :] + bias.abs()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        got = out[rows].float()
        assert torch.isfinite(got).all()
        tol = 1e-4 * mag + (0 if epi == 2 else 2.0 ** -8 * ref.abs()) + 1e-6
        bad = (got - ref).abs() > tol
        assert not bad.any(), f"rows {r0}: {int(bad.sum())} beyond tolerance, max |d| {(got - ref).abs().max().item():.3e}"


def test_layernorm_q8_matches_torch(lib):
    """LayerNorm -> per-row e4m3 (rr_op_layernorm_q8): scales = row amax / 448 of the fp32 LayerNorm output, codes equal
    torch's e4m3 rounding of the same values except where a last-bit difference of the fp32 LayerNorm arithmetic sits on a
    rounding boundary (then they differ by one code step)."""
    rows, cols = 777, 1024
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(rows, cols, generator=g) * 1.3 + 0.4).cuda()
    x[3] = 0.0                                                            # zero variance row: LN output = beta
    gamma, beta = (1 + 0.2 * torch.randn(cols, generator=g)).cuda(), (0.1 * torch.randn(cols, generator=g)).cuda()
    out = torch.zeros(rows, cols, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(rows, device="cuda")
    st = torch.zeros(rows, 2, device="cuda")
    eps = 1e-12
    assert lib.rr_op_layernorm_q8(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, rows, cols, out.data_ptr(), sc.data_ptr(),
                                  st.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    y = torch.nn.functional.layer_norm(x, (cols,), gamma, beta, eps)
    want_s = y.abs().amax(1) / 448.0
    assert torch.allclose(sc, want_s, rtol=3e-6, atol=0)
    deq = out.view(torch.float8_e4m3fn).float() * sc[:, None]
    # e4m3 has 3 mantissa bits: |dequantised - y| <= half a code step (2^-4 relative) + the scale's share of the subnormal step
    assert ((deq - y).abs() <= 0.0626 * y.abs() + sc[:, None] * 2.0 ** -10 + 1e-7).all()
    want = (y / sc[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    frac = (out != want).float().mean().item()
    assert frac < 2e-3, f"{frac:.2e} of the codes differ from torch's rounding of torch's LayerNorm"
    assert torch.allclose(st[:, 0], x.mean(1), atol=1e-5) and torch.allclose(st[4:, 1], 1 / torch.sqrt(x[4:].var(1, unbiased=False) + eps), rtol=1e-4)


def _e4m3_operands(M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g) * (0.2 + 3.0 * torch.rand(M, 1, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g) * (0.01 + 0.08 * torch.rand(N, 1, device="cuda", generator=g))
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    sa, sw = a.abs().amax(1) / 448.0, w.abs().amax(1) / 448.0
    return (a / sa[:, None]).to(torch.float8_e4m3fn), (w / sw[:, None]).to(torch.float8_e4m3fn), bias, sa, sw, g


@pytest.mark.parametrize("M,N,K", [(8192, 4096, 256), (8300, 4096, 128), (33_000, 1024, 384)])
def test_gemm_fp8_gelu_to_e4m3_epilogue(lib, M, N, K):
    """rr_op_gemm_fp8_gelu_e4m3 (gemm_kernel_hp8 EPI 3): the FFN-up of the fp8 configuration writing its GELU output as e4m3
    bytes under one static scale.  Against torch: the fp32 pre-activation of the same e4m3 operands, exact erf-GELU, x out_mul,
    clamp, torch's own e4m3 rounding — byte for byte except where the two fp32 values straddle a rounding boundary (different
    summation order, the 4.8e-7 GELU polynomial): those may differ by ONE e4m3 step and must be rare."""
    a8, w8, bias, sa, sw, _ = _e4m3_operands(M, N, K, M + N + K)
    out = torch.full((M, N), 0x7F, dtype=torch.uint8, device="cuda")
    mul = 8.0
    assert lib.rr_op_gemm_fp8_gelu_e4m3(a8.view(torch.uint8).data_ptr(), w8.view(torch.uint8).data_ptr(), bias.data_ptr(), sa.data_ptr(),
                                        sw.data_ptr(), mul, M, N, K, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    for r0 in sorted({0, max(0, M // 2 - 150), max(0, M - 300)}):
        rows = slice(r0, min(M, r0 + 300))
        pre = (a8[rows].float() @ w8.float().T) * sa[rows, None] * sw[None, :] + bias
        want = (torch.nn.functional.gelu(pre) * mul).clamp(-448, 448).to(torch.float8_e4m3fn)
        got = out[rows].view(torch.float8_e4m3fn)
        differ = got.view(torch.uint8) != want.view(torch.uint8)
        frac = differ.float().mean().item()
        wf, gf = want.float(), got.float()
        step = torch.maximum(wf.abs(), gf.abs()) * 2.0 ** -3 + 2.0 ** -9       # one e4m3 step (3 mantissa bits; subnormal spacing 2^-9)
        assert torch.isfinite(gf).all()
        assert ((gf - wf).abs() <= step)[differ].all(), "a byte is more than one e4m3 step away"
        assert frac < 2e-3, f"rows {r0}: {frac:.2e} of the bytes differ"


@pytest.mark.parametrize("M,N,K,with_stats", [(33_000, 1024, 512, True), (33_100, 1024, 4096, True), (66_000, 768, 256, False)])
def test_gemm_fp8_residual_epilogue(lib, M, N, K, with_stats):
    """rr_op_gemm_fp8_resid (gemm_kernel_hp8 EPI 4): out = scale * col_scale * (A8 . W8^T) + bias + residual row, the residual
    either given or recomputed from (mean, rstd), gamma, beta — the FFN-down of the fp8 configuration."""
    a8, w8, bias, _, sw, g = _e4m3_operands(M, N, K, M + N + K + 1)
    scale = 0.125
    x = torch.randn(M, N, device="cuda", generator=g) * 2.0 + 0.3
    gamma = 1.0 + 0.1 * torch.randn(N, device="cuda", generator=g)
    beta = 0.05 * torch.randn(N, device="cuda", generator=g)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    stats = torch.stack([mean, torch.rsqrt(var + 1e-12)], 1).contiguous()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    assert lib.rr_op_gemm_fp8_resid(a8.view(torch.uint8).data_ptr(), w8.view(torch.uint8).data_ptr(), bias.data_ptr(), scale, sw.data_ptr(),
                                    x.data_ptr(), stats.data_ptr() if with_stats else None, gamma.data_ptr() if with_stats else None,
                                    beta.data_ptr() if with_stats else None, M, N, K, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    for r0 in sorted({0, max(0, M // 2 - 150), max(0, M - 300)}):
        rows = slice(r0, min(M, r0 + 300))
        af, wf = a8[rows].float(), w8.float()
        res = ((x[rows] - mean[rows, None]) * stats[rows, 1:2] * gamma + beta) if with_stats else x[rows]
        ref = (af @ wf.T) * scale * sw[None, :] + bias + res
        mag = (af.abs() @ wf.abs().T) * scale * sw[None, :] + bias.abs() + res.abs()
        got = out[rows]
        assert torch.isfinite(got).all()
        bad = (got - ref).abs() > 1e-4 * mag + 1e-6
        assert not bad.any(), f"rows {r0}: {int(bad.sum())} beyond tolerance, max |d| {(got - ref).abs().max().item():.3e}"
    # the ring only: shapes below 512 tiles are refused, not silently run elsewhere
    assert lib.rr_op_gemm_fp8_resid(a8.view(torch.uint8).data_ptr(), w8.view(torch.uint8).data_ptr(), bias.data_ptr(), scale, sw.data_ptr(),
                                    x.data_ptr(), None, None, None, 1024, N, K, out.data_ptr(), _stream()) != 0


def test_fp8_ffn_down_forward_against_the_oracle_emulation():
    """The fp8 configuration with BOTH FFN GEMMs on the e4m3 ring (rows large enough for the persistent kernel: 2 048 pairs of
    64 tokens on a 3-layer 256-wide model): device forward against the oracle with the same rounding points
    (device_rounding(fp8=True, fp8_down=True): GELU output e4m3 under the static scale 8, FFN-down weights per channel), and
    against the same model with the 16-bit FFN-down (rr_set_tuning fp8_ffn_down 0)."""
    import rmr_amd
    from helpers import O, arch_from_cfg, record_margin
    from rmr_amd import _lib
    lib_ = _lib.load()
    cfg = O.OracleConfig(vocab_size=2000, hidden=256, layers=3, heads=4, intermediate=1024, max_pos=64, ce_hidden=256,
                         ce_heads=4, ce_intermediate=1024, ce_layers=2, ce_max_pos=128, li_dim=64)
    cfg.loss_fn = "BCE"
    Bq, K, S = 8, 256, 64
    w = O.make_weights(cfg, seed=2, vision=False)
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=4)
    arch = arch_from_cfg(cfg, False, "fp16")
    arch["fp8"] = 1
    eng = rmr_amd.RerankEngine(arch)
    eng.set_option("fp8_first_layer", 0)         # the whole-stack e4m3 form (the shipped default is the last layer: see the ranking tests)
    eng.load_state_dict(w)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K)
    eng.set_option("fp8_ffn_down", 1)            # opt-in (a handle option; the default keeps FFN-down in 16 bits)
    got = eng.forward_ids(*args)["logits"].cpu()
    eng.set_option("fp8_ffn_down", 0)
    up_only = eng.forward_ids(*args)["logits"].cpu()
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K).logits.reshape(-1)
        with O.device_rounding(torch.float16, fp8=True, fp8_down=True) as mm:
            emu = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(-1)
    d_emu, d_ref, e_ref, u_ref = ((got - emu).abs().max().item(), (got - ref).abs().max().item(), (emu - ref).abs().max().item(),
                                  (up_only - ref).abs().max().item())
    print(f"[fp8 ffn-down] device vs same-rounding oracle {d_emu:.2e}; device vs fp32 {d_ref:.2e} (16-bit FFN-down: {u_ref:.2e}); "
          f"oracle emulation vs fp32 {e_ref:.2e}; the switch moves the logits by {(got - up_only).abs().max().item():.2e}")
    record_margin("fp8_small_ffn_down/fp16", vs_emulation=d_emu, vs_fp32=d_ref, ffn_down_16bit_vs_fp32=u_ref, emulation_vs_fp32=e_ref)
    assert torch.isfinite(got).all()
    assert (got - up_only).abs().max().item() > 0          # the e4m3 FFN-down really ran
    # same rounding points, but 2 048 logits each behind ~1e6 e4m3 roundings that a 1e-7 perturbation can flip (6 % of an
    # element per flip): the maximum over the batch sits near the drift itself (measured 4.1e-3 against 7.0e-3)
    assert d_emu <= max(2e-3, 0.8 * e_ref)
    assert d_ref <= 2.0 * e_ref + 1e-3


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_fp8_forward_small_model_against_the_oracle_emulation(dt):
    """rr_config.fp8 on a 3-layer 256-wide model (K = 256: two K-tiles; small problems run the two-stage e4m3 kernel):
    the device forward against the oracle with the same e4m3 rounding points (device_rounding(fp8=True)), and the drift
    both have from the fp32 forward."""
    import rmr_amd
    from helpers import O, arch_from_cfg
    cfg = O.OracleConfig(vocab_size=2000, hidden=256, layers=3, heads=4, intermediate=1024, max_pos=64, ce_hidden=256,
                         ce_heads=4, ce_intermediate=1024, ce_layers=2, ce_max_pos=128, li_dim=64)
    cfg.loss_fn = "BCE"
    Bq, K, S = 2, 6, 64
    w = O.make_weights(cfg, seed=2, vision=False)
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=4)
    arch = arch_from_cfg(cfg, False, dt)
    arch["fp8"] = 1
    eng = rmr_amd.RerankEngine(arch)
    eng.set_option("fp8_first_layer", 0)         # the whole-stack e4m3 form (the shipped default is the last layer: see the ranking tests)
    eng.load_state_dict(w)
    r = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), Bq, K, want_order=True)
    torch.cuda.synchronize()
    got = r["logits"].cpu()
    with torch.no_grad():
        ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K).logits.reshape(-1)
        with O.device_rounding(torch.float16 if dt == "fp16" else torch.bfloat16, fp8=True) as mm:
            emu = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(-1)
        with O.device_rounding(torch.float16 if dt == "fp16" else torch.bfloat16, fold=False) as mm:
            b16 = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(-1)
    d_emu, d_ref, e_ref = (got - emu).abs().max().item(), (got - ref).abs().max().item(), (emu - ref).abs().max().item()
    print(f"[fp8 small/{dt}] device vs same-rounding oracle {d_emu:.2e}; device vs fp32 {d_ref:.2e}; oracle-fp8 vs fp32 {e_ref:.2e}; "
          f"16-bit vs fp32 {(b16 - ref).abs().max().item():.2e}")
    assert torch.isfinite(got).all()
    assert d_emu <= max(2e-3, 0.5 * e_ref)          # same rounding points: well inside the e4m3 drift itself
    assert d_ref <= 2.0 * e_ref + 1e-3
    assert r["order"].cpu().tolist() == [O.rank_descending_stable(x) for x in got.view(Bq, K).tolist()]
    # the subset options ("fp8_first_layer", "fp8_qkv"): layers below the first e4m3 layer run the folded 16-bit dataflow, the first
    # e4m3 layer reads its predecessor's raw rows through the folded QKV — against the oracle with the same choices
    for first, qkv in ((1, 1), (2, 1), (3, 1), (0, 0), (1, 0)):
        eng.set_option("fp8_first_layer", first)
        eng.set_option("fp8_qkv", qkv)
        got_s = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), Bq, K)["logits"].cpu()
        with torch.no_grad(), O.device_rounding(torch.float16 if dt == "fp16" else torch.bfloat16, fp8=True, fp8_first=first, fp8_qkv=bool(qkv)) as mm:
            emu_s = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(-1)
        d_s, e_s = (got_s - emu_s).abs().max().item(), (emu_s - ref).abs().max().item()
        print(f"[fp8 small/{dt}] first e4m3 layer {first}, e4m3 QKV {qkv}: device vs same-choice oracle {d_s:.2e}; that oracle vs fp32 {e_s:.2e}")
        assert torch.isfinite(got_s).all() and d_s <= max(2e-3, 0.5 * e_s)


# Gates of the e4m3 drift on c5_full for the WHOLE-STACK form ("fp8_first_layer" = 0), FROZEN at absolute numbers since round 4
# (they used to be 2 x whatever was measured).  e4m3 QKV / FFN-up, 16-bit FFN-down: measured 4.6e-2 abs / 3.9e-2
# centred, rank correlation 0.970.  Opt-in "fp8_ffn_down" (GELU output as e4m3 under the static scale 8): 7.8-8.2e-2 / 4.5-4.9e-2,
# 0.951-0.964.  Whether a configuration RANKS like the fp32 reference is what the c5 ranking fixtures decide (the tests below).
FP8_GATES = {0: dict(abs=0.06, centred=0.05, rho=0.95), 1: dict(abs=0.10, centred=0.06, rho=0.93)}


def _fp8_engine(name):
    import rmr_amd
    from helpers import arch_from_cfg, load_fullsize
    cfg, w, vision, qs = load_fullsize(name)
    arch = arch_from_cfg(cfg, vision, "fp16")
    arch["fp8"] = 1
    eng = rmr_amd.RerankEngine(arch)
    eng.set_option("fp8_first_layer", 0)         # the whole-stack e4m3 form (the shipped default is the last layer: see the ranking tests)
    eng.load_state_dict(w)
    return eng, qs


def test_fp8_forward_bert_large_against_the_c5_golden():
    """BASELINE configs[4]: bert-large, K = 200, S = 512, e4m3 QKV / FFN-up GEMMs (and, opt-in, FFN-down).  north_star's 1e-3 is a
    16-bit figure; for e4m3 the drift from the committed fp32 stock-HF logits is REPORTED and bounded by frozen gates: absolute,
    centred per candidate list (what ranking sees) and rank correlation, next to the reference's own bf16-autocast drift."""
    from helpers import margin_stats, record_margin
    eng, qs = _fp8_engine("c5_full")
    q = qs[0]
    K = q["ids"].shape[0]
    ref, ac = q["fp32"], q["autocast"]
    for down in (0, 1):
        eng.set_option("fp8_ffn_down", down)
        r = eng.forward_ids(q["ids"].cuda(), q["am"].cuda(), q["tt"].cuda(), 1, K, want_order=True)
        torch.cuda.synchronize()
        got = r["logits"].cpu()
        st = margin_stats(got, ref)
        gate = FP8_GATES[down]
        print(f"[c5_full/fp8+fp16, fp8_ffn_down={down}] K={K}: |dlogit| vs fp32 max {st['max_abs']:.3e}, centred {st['centred']:.3e}; "
              f"logit std over the list {ref.std():.3f}; rank correlation {st['rho']:.4f}; top-5 overlap {st['top5']}; "
              f"reference bf16-autocast drift {(ac - ref).abs().max():.3e}")
        keys = ["c5_full/fp8+fp16", "c5_full/fp8+fp16/ffn_down_16bit"] if down == 0 else ["c5_full/fp8+fp16/ffn_down_e4m3"]
        for k in keys:
            record_margin(k, gate_abs=gate["abs"], gate_centred=gate["centred"], gate_rho=gate["rho"], fp8_ffn_down=down,
                          reference_bf16_autocast_drift=float((ac - ref).abs().max()), **st)
        assert torch.isfinite(got).all()
        assert st["max_abs"] <= gate["abs"] and st["centred"] <= gate["centred"] and st["rho"] >= gate["rho"]


def _rank_lists(eng, name, qs, tag):
    """One forward per query of a ranking fixture: per query dict(stats vs fp32, top-5 set kept, yardstick), the engine's and the fp32
    reference's Recall@5/10; every margin recorded under `<fixture>/q<i>/<tag>` next to the reference's own autocast figures."""
    import rmr_amd
    from helpers import O, margin_stats, ranking_yardstick, record_margin, top5_set
    ranked, ranked_ref, pos, rows = [], [], [], []
    for qi, q in enumerate(qs):
        y = ranking_yardstick(q)
        sel, ref = y["sel"], y["ref"]
        r = eng.forward_ids(q["ids"][sel].cuda(), q["am"][sel].cuda(), q["tt"][sel].cuda(), 1, len(sel), want_order=True)
        torch.cuda.synchronize()
        lg = r["logits"].cpu()
        order, ref_order = r["order"][0].cpu().tolist(), O.rank_descending_stable(ref.tolist())
        st = margin_stats(lg, ref)
        kept = set(order[:5]) == set(ref_order[:5])
        print(f"[{name}/{tag} q{qi}] |dlogit| {st['max_abs']:.3e} centred {st['centred']:.3e} rho {st['rho']:.4f} top-5 {st['top5']} "
              f"{'kept' if kept else 'LOST'}; gap {y['gap']:.3f}; reference bf16-autocast: |d| {y['stats']['max_abs']:.3e} centred "
              f"{y['stats']['centred']:.3e} top-5 {'kept' if y['autocast_keeps_top5'] else 'lost'}; rule {'binds' if y['binds'] else 'does not bind'}")
        record_margin(f"{name}/q{qi}/{tag}", gap_5_6=y["gap"], top5_set_kept=bool(kept), rule_binds=y["binds"],
                      reference_autocast_pool_max_abs=y["pool_autocast_max_abs"], reference_autocast_max_abs=y["stats"]["max_abs"], reference_autocast_centred=y["stats"]["centred"],
                      reference_autocast_keeps_top5=y["autocast_keeps_top5"], **st)
        assert torch.isfinite(lg).all() and order == O.rank_descending_stable(lg.tolist())
        assert top5_set(lg) == set(order[:5])
        ranked.append(order); ranked_ref.append(ref_order); pos.append([int(q["positive_list_index"])])
        rows.append(dict(stats=st, kept=kept, y=y))
    return rows, rmr_amd.recall_precision_at_k(ranked, pos, [5, 10]), O.recall_precision_at_k(ranked_ref, pos, [5, 10])


def _fixtures():
    import os
    from helpers import GOLDEN, RANKING_FIXTURES_C5
    return [n for n in RANKING_FIXTURES_C5 if os.path.exists(os.path.join(GOLDEN, f"{n}.npz"))]


@pytest.mark.parametrize("name", _fixtures())
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_16_bit_ranking_on_the_c5_ranking_fixtures(name, dt):
    """The c5 ranking fixtures (tests/golden/make_golden.py SEP: bert-large, two queries x 200 candidates chosen from a pool of 300
    so that the fp32 stock-HF logits leave a designed gap between rank 5 and rank 6; gains 2.5 (c5_sep, gap 0.12; c5_sep_wide, the
    widest gap the pool allows), 2.0, 1.5), each carrying the logits of the REFERENCE's own arithmetic (bf16 autocast) on the same
    lists.  Gates, every one derived from that yardstick and the designed gap, none from the value observed:
      * bf16 (the reference's own operand type): max |d| <= 1.5 x the autocast reference's max |d| over the query's candidate POOL
        (the figure the fixture was generated with; the rule of helpers.bf16_gate: one draw of the same rounding noise against
        another — the maximum over the 200 selected candidates alone is too noisy a yardstick: on c5_sep_wide q1 it is 0.082 where
        the pool's is 0.104), and the fp32 top-5 set is kept on every list where
        the rule binds (autocast keeps it with max |d| <= gap / 4: then 1.5 x that is < gap / 2, so the drift bound itself implies
        it); on the other lists (the autocast reference's own drift is 0.6 - 0.9 of the gap there: it keeps its top-5 by the draw,
        not by margin) the outcome is recorded, not gated;
      * fp16 (3 more mantissa bits = 8 x finer operand rounding): max |d| <= 1/4 of the same pool figure, and the top-5 set
        kept wherever the autocast reference keeps it — then Recall@5/10 equal the fp32 reference's (1/0 and 1/1 by construction)."""
    import rmr_amd
    from helpers import arch_from_cfg, load_fullsize
    cfg, w, vision, qs = load_fullsize(name)
    eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, vision, dt))
    eng.load_state_dict(w)
    rows, got, want = _rank_lists(eng, name, qs, dt)
    assert want["recall"] == [0.5, 1.0]
    for r in rows:
        ac = r["y"]["pool_autocast_max_abs"]
        assert r["stats"]["max_abs"] <= (1.5 if dt == "bf16" else 0.25) * ac
        if r["y"]["binds"] if dt == "bf16" else r["y"]["autocast_keeps_top5"]:
            assert r["kept"]
    if all(r["y"]["binds"] if dt == "bf16" else r["y"]["autocast_keeps_top5"] for r in rows):
        assert got["recall"][0] == want["recall"][0]      # Recall@5: what the designed gap protects (nothing separates rank 10 from 11)


# The e4m3 configuration on the ranking fixtures.  What the device study found (tests/tools/fp8_subset_study.py,
# profiles/r05_*_fp8_subset_study.json): with e4m3 QKV / FFN-up in EVERY layer a widened random bert-large does not keep the fp32
# top-5 where the reference's own autocast arithmetic does; the drift grows with the number of e4m3 layers — one layer (the last):
# about the autocast reference's own drift; two: 1.5 - 2 x, INSIDE or OUTSIDE half the gap depending on how unrelated roundings fall
# (a GELU polynomial of another degree moved c5_sep_wide q0 from 0.137 to 0.193 against a half-gap of 0.18); all 24: 3 - 8 x, rank
# correlation 0.4 - 0.9.  The subset that ranks with a margin worth the name is therefore the LAST layer: the default of rr_config.fp8
# (handle option "fp8_first_layer" = layers - 1); the whole-stack form is an opt-in whose verdict is frozen below.  "Ranks with
# margin" := on every fixture where the rule binds (helpers.ranking_yardstick) the fp32 top-5 set is kept AND the centred drift (what
# a ranking sees) is <= half the designed rank-5/6 gap.
FP8_WHOLE_STACK_RANKS = False        # frozen verdict of "fp8_first_layer" = 0 (both FFN-down forms); a change that flips it must edit this line
FP8_WHOLE_STACK_SANITY = dict(max_abs=1.2, rho=0.3)     # frozen: finite, bounded, still correlated with the reference (measured <= 0.82 / >= 0.375)


def _ranks_with_margin(rows):
    binding = [r for r in rows if r["y"]["binds"]]
    return binding, all(r["kept"] and r["stats"]["centred"] <= 0.5 * r["y"]["gap"] for r in binding)


@pytest.mark.parametrize("name", _fixtures())
def test_fp8_default_subset_ranks_wherever_the_reference_arithmetic_does(name):
    """rr_config.fp8 as shipped (e4m3 QKV / FFN-up in the last text-encoder layer): on every list where the rule binds the fp32
    top-5 set is kept with the centred drift inside half the designed gap, and Recall@5 equals the fp32 reference's."""
    import rmr_amd
    from helpers import arch_from_cfg, load_fullsize
    cfg, w, vision, qs = load_fullsize(name)
    arch = arch_from_cfg(cfg, vision, "fp16")
    arch["fp8"] = 1
    eng = rmr_amd.RerankEngine(arch)
    eng.load_state_dict(w)
    assert eng.get_option("fp8_first_layer") == cfg.layers - 1 and eng.get_option("fp8_qkv") == 1 and eng.get_option("fp8_ffn_down") == 0
    rows, got, want = _rank_lists(eng, name, qs, "fp8/default_last1")
    binding, ok = _ranks_with_margin(rows)
    assert ok
    if len(binding) == len(rows):
        assert got["recall"][0] == want["recall"][0]      # Recall@5: what the designed gap protects (nothing separates rank 10 from 11)


@pytest.mark.parametrize("down", [0, 1])
def test_fp8_whole_stack_verdict_is_the_frozen_one(down):
    """"fp8_first_layer" = 0 (e4m3 in all 24 layers; with and without the e4m3 FFN-down): the ranking verdict over ALL c5 ranking
    fixtures is the frozen FP8_WHOLE_STACK_RANKS — an equality, so that a change which makes the whole stack rank (or one that breaks
    a fixture) fails here and has to edit the constant and the documentation — and the drift stays inside frozen sanity bounds."""
    import rmr_amd
    from helpers import arch_from_cfg, load_fullsize
    all_rows = []
    for name in _fixtures():
        cfg, w, vision, qs = load_fullsize(name)
        arch = arch_from_cfg(cfg, vision, "fp16")
        arch["fp8"] = 1
        eng = rmr_amd.RerankEngine(arch)
        eng.load_state_dict(w)
        eng.set_option("fp8_first_layer", 0)
        eng.set_option("fp8_ffn_down", down)
        rows, _, _ = _rank_lists(eng, name, qs, "fp8/whole_stack/ffn_down_" + ("e4m3" if down else "16bit"))
        all_rows += rows
        del eng
        torch.cuda.empty_cache()
    for r in all_rows:
        assert r["stats"]["max_abs"] <= FP8_WHOLE_STACK_SANITY["max_abs"] and r["stats"]["rho"] >= FP8_WHOLE_STACK_SANITY["rho"]
    binding, ok = _ranks_with_margin(all_rows)
    assert binding, "no fixture binds: the verdict would be vacuous"
    assert ok == FP8_WHOLE_STACK_RANKS
