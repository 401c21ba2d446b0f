"""GPU parity of the stand-alone HIP operators, called through the C ABI (rr_op_*), against plain fp32
torch references on the same (bf16-rounded) inputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import rmr_amd
    from rmr_amd import _lib
    return _lib.load()       # raises if librerank_mi355.so is missing: no fallback


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _bits(t):
    return t.view(torch.int16)


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 768, 768), (200, 2304, 768), (77, 128, 768),
                                   (1, 2048, 768), (333, 768, 3072), (130, 64, 128), (512, 3072, 768)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(lib, M, N, K, epi):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.7).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    rc = lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + b
    ref = {0: ref, 1: _gelu(ref), 2: ref, 3: torch.tanh(ref)}[epi]
    got = out.float()
    assert torch.isfinite(got).all()
    tol = 2e-4 if epi == 2 else 1.2e-2     # fp32 out: accumulation-order noise; bf16 out: half-ulp of |x| <~ 3
    err = (got - ref).abs()
    assert (err <= tol * (1 + ref.abs())).all(), f"max err {err.max().item()}"


def test_gemm_no_bias_and_resid(lib):
    M, N, K = 300, 768, 768
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), 0, M, N, K, 2, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert torch.allclose(out, A.float() @ W.float().t(), atol=3e-4, rtol=1e-4)
    assert lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K,
                                    out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert torch.allclose(out, A.float() @ W.float().t() + b + R, atol=3e-4, rtol=1e-4)


def test_gemm_asymmetric_identity(lib):
    """A = I against an asymmetric W catches a transposed C-write (cdna guide §3)."""
    K = 128
    A = torch.eye(K).bfloat16().cuda()
    W = (torch.arange(K * K, dtype=torch.float32).view(K, K) % 251 - 125).bfloat16().cuda()   # exact in bf16
    out = torch.empty(K, K, device="cuda")
    assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), 0, K, K, K, 2, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, W.float().t())


def test_gemm_rejects_bad_shapes(lib):
    A = torch.zeros(8, 96).bfloat16().cuda()
    out = torch.zeros(8, 8, device="cuda")
    assert lib.rr_op_gemm_bf16(A.data_ptr(), A.data_ptr(), 0, 8, 8, 96, 2, out.data_ptr(), _stream()) == -2   # K % 64


def _attn_ref(q, k, v, bias, heads, qdiv=1):
    B, Tk = k.shape[0], k.shape[1]
    Tq = q.shape[1]
    qq = q.float().repeat_interleave(qdiv, 0)[:B]
    qh = qq.view(B, Tq, heads, 64).transpose(1, 2)
    kh = k.float().view(B, Tk, heads, 64).transpose(1, 2)
    vh = v.float().view(B, Tk, heads, 64).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * math.log(2.0)   # q arrives pre-scaled by log2(e)/sqrt(dh): the kernel's exponentials are base 2
    if bias is not None:
        s = s + bias[:, None, None, :]
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Tq, heads * 64)


@pytest.mark.parametrize("B,heads,Tq,Tk,masked", [(2, 2, 64, 64, False), (3, 12, 128, 128, True), (2, 12, 512, 512, True),
                                                   (2, 2, 593, 593, True), (4, 12, 49, 49, False), (6, 12, 49, 32, False),
                                                   (1, 1, 1, 1, False), (2, 3, 70, 130, True)])
def test_attention(lib, B, heads, Tq, Tk, masked):
    g = torch.Generator().manual_seed(B * 100 + Tq + Tk)
    H = heads * 64
    q = (torch.randn(B, Tq, H, generator=g) * 0.5).bfloat16().cuda()
    k = torch.randn(B, Tk, H, generator=g).bfloat16().cuda()
    v = torch.randn(B, Tk, H, generator=g).bfloat16().cuda()
    bias = None
    if masked:
        keep = torch.rand(B, Tk, generator=g) > 0.3
        keep[:, 0] = True
        if Tk > 40:
            keep[0, 5:40] = False                       # a long masked run inside a tile
        bias = torch.where(keep, 0.0, -1e30).float().cuda()
    out = torch.full((B, Tq, H), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = lib.rr_op_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), H, H, bias.data_ptr() if masked else 0,
                                  B, heads, Tq, Tk, 1, out.data_ptr(), H, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref = _attn_ref(q, k, v, bias, heads)
    got = out.float()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 2e-2       # bf16 P and bf16 output on |O| <~ 1


def test_attention_fused_qkv_layout_and_query_broadcast(lib):
    """strided q/k/v inside one [rows, 3H] buffer (as the QKV GEMM writes it) and per-query Q shared by K pairs."""
    B, heads, T = 3, 2, 96
    H = heads * 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * T, 3 * H, generator=g).bfloat16().cuda()
    out = torch.empty(B, T, H, dtype=torch.bfloat16, device="cuda")
    base = qkv.data_ptr()
    rc = lib.rr_op_attention_bf16(base, base + 2 * H, base + 4 * H, 3 * H, 3 * H, 0, B, heads, T, T, 1,
                                  out.data_ptr(), H, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    x = qkv.view(B, T, 3 * H)
    ref = _attn_ref(x[..., :H], x[..., H:2 * H], x[..., 2 * H:], None, heads)
    assert (out.float() - ref).abs().max().item() < 2e-2
    # q_batch_div: 2 queries, each shared by 3 pairs
    q = torch.randn(2, 49, H, generator=g).bfloat16().cuda()
    k = torch.randn(6, 32, H, generator=g).bfloat16().cuda()
    v = torch.randn(6, 32, H, generator=g).bfloat16().cuda()
    out = torch.empty(6, 49, H, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), H, H, 0, 6, heads, 49, 32, 3,
                                    out.data_ptr(), H, _stream()) == 0
    torch.cuda.synchronize()
    ref = _attn_ref(q, k, v, None, heads, qdiv=3)
    assert (out.float() - ref).abs().max().item() < 2e-2


def test_attention_all_masked_row_is_uniform(lib):
    B, heads, T = 2, 1, 100
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, T, 64, generator=g).bfloat16().cuda()
    k = torch.randn(B, T, 64, generator=g).bfloat16().cuda()
    v = torch.randn(B, T, 64, generator=g).bfloat16().cuda()
    bias = torch.zeros(B, T)
    bias[1] = -1e30                                     # pair 1: no valid key at all
    bias = bias.cuda()
    out = torch.empty(B, T, 64, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), 64, 64, bias.data_ptr(), B, heads, T, T,
                                    1, out.data_ptr(), 64, _stream()) == 0
    torch.cuda.synchronize()
    want = v[1].float().mean(0, keepdim=True).expand(T, 64)
    assert (out[1].float() - want).abs().max().item() < 1e-2
    ref0 = _attn_ref(q[:1], k[:1], v[:1], None, 1)
    assert (out[0].float() - ref0[0]).abs().max().item() < 2e-2


def test_attention_online_softmax_rescale(lib):
    """Force the running max to jump in a late tile (rule 26: rare data-dependent branch needs its own test)."""
    B, heads, T = 1, 1, 256
    g = torch.Generator().manual_seed(4)
    q = torch.randn(B, T, 64, generator=g) * 0.3
    k = torch.randn(B, T, 64, generator=g) * 0.3
    k[0, 200] = q[0, 17] * 40.0                         # key 200 (tile 3) spikes against query 17
    k[0, 70] = q[0, 90] * 25.0
    v = torch.randn(B, T, 64, generator=g)
    q, k, v = q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda()
    out = torch.empty(B, T, 64, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), 64, 64, 0, B, heads, T, T, 1,
                                    out.data_ptr(), 64, _stream()) == 0
    torch.cuda.synchronize()
    ref = _attn_ref(q, k, v, None, 1)
    assert (out.float() - ref).abs().max().item() < 3e-2


def _run_attn(lib, q, k, v, bias, heads):
    B, Tq, H = q.shape
    out = torch.full((B, Tq, H), float("nan"), dtype=q.dtype, device="cuda")
    assert lib.rr_op_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), H, H, bias.data_ptr() if bias is not None else 0,
                                    B, heads, Tq, k.shape[1], 1, out.data_ptr(), H, _stream()) == 0
    torch.cuda.synchronize()
    return out.float()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("dt", [0, 1])
def test_attention_fixed_reference_schedule(lib, dt, mode):
    """The default schedule (two launches, grids >= 1024 workgroups) takes the row maximum of the first tile with a valid
    key as a fixed softmax reference; workgroups where a row sum leaves 2^64, or that see no valid key at all, are
    recomputed in the online form by the second launch.  Cases: (a) a later tile 2^20 above the reference stays on
    the fixed path; (b) 2^230 above it overflows and is recomputed; (c) left padding (tile 0 fully masked) takes its
    reference from tile 1; (d) a pair with no valid key is recomputed and comes out uniform.  All against the fp32
    reference and against the online-only schedule of the same kernel.  dt = 1: fp16 operands, where P itself must stay
    below 65504, so (a) is recomputed as well."""
    t16 = torch.float16 if dt else torch.bfloat16
    assert lib.rr_set_op_dtype(dt) == 0
    try:
        _fixed_reference_cases(lib, t16, mode)
    finally:
        lib.rr_set_op_dtype(0)


def _fixed_reference_cases(lib, t16, mode):
    """mode 1: 32 query rows per wave (128-row workgroups); mode 2: 64 rows per wave (256-row workgroups)."""
    B, heads, T = 48, 8, 320                          # 1 152 workgroups: the two-launch schedule starts at 1 024
    H = heads * 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, T, H, generator=g) * 0.3
    k = torch.randn(B, T, H, generator=g) * 0.3
    v = torch.randn(B, T, H, generator=g)
    k[0, 200, :64] = q[0, 17, :64] * (20.0 / float((q[0, 17, :64] ** 2).sum()))      # (a) head 0, query 17: +20
    k[1, 300, 64:128] = q[1, 5, 64:128] * (230.0 / float((q[1, 5, 64:128] ** 2).sum()))   # (b) head 1, query 5: +230
    bias = torch.zeros(B, T)
    bias[2, :70] = -1e30                                                            # (c) tile 0 and 6 keys of tile 1
    bias[0, 250:] = -1e30                                                           # ordinary tail padding on (a)
    bias[3, :] = -1e30                                                              # (d) no valid key
    bias[4:, 300:] = -1e30
    q, k, v, bias = q.to(t16).cuda(), k.to(t16).cuda(), v.to(t16).cuda(), bias.cuda()
    ref = _attn_ref(q, k, v, bias, heads)
    try:
        assert lib.rr_set_tuning(b"attn_fixed_ref", mode) == 0
        got = _run_attn(lib, q, k, v, bias, heads)
        assert lib.rr_set_tuning(b"attn_fixed_ref", 0) == 0
        online = _run_attn(lib, q, k, v, bias, heads)
    finally:
        lib.rr_set_tuning(b"attn_fixed_ref", -1)          # back to the built-in default
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 3e-2
    assert (online - ref).abs().max().item() < 3e-2
    assert (got - online).abs().max().item() < 2e-2
    # the recomputation IS the online form: (b) the 128-query block of head 1 that holds query 5, (d) every block of pair 3
    assert torch.equal(got[1, :128, 64:128], online[1, :128, 64:128]) and torch.equal(got[3], online[3])
    assert (got[3] - v[3].float().mean(0, keepdim=True)).abs().max().item() < 1e-2  # uniform attention
    assert not torch.equal(got[4:], online[4:])                                     # ... and the rest did take the fixed path


@pytest.mark.parametrize("mode", [1, 2])
def test_attention_every_workgroup_recomputed(lib, mode):
    """No valid key anywhere: the first launch flags every workgroup and the second one recomputes all of them."""
    B, heads, T = 48, 8, 384                                # 48 * 8 * 3 = 1 152 workgroups of 128 rows
    H = heads * 64
    g = torch.Generator().manual_seed(13)
    q = torch.randn(B, T, H, generator=g).bfloat16().cuda()
    k = torch.randn(B, T, H, generator=g).bfloat16().cuda()
    v = torch.randn(B, T, H, generator=g).bfloat16().cuda()
    bias = torch.full((B, T), -1e30).cuda()
    try:
        assert lib.rr_set_tuning(b"attn_fixed_ref", mode) == 0
        got = _run_attn(lib, q, k, v, bias, heads)
    finally:
        lib.rr_set_tuning(b"attn_fixed_ref", -1)
    want = v.float().mean(1, keepdim=True).expand(B, T, H)
    assert torch.isfinite(got).all() and (got - want).abs().max().item() < 1e-2


def test_attention_fixed_reference_forms_are_bitwise_equal(lib):
    """32 and 64 query rows per wave do the same arithmetic per row (same tile order, same reference, same skips)."""
    B, heads, T = 24, 12, 512
    H = heads * 64
    g = torch.Generator().manual_seed(12)
    q = (torch.randn(B, T, H, generator=g) * 0.4).bfloat16().cuda()
    k = torch.randn(B, T, H, generator=g).bfloat16().cuda()
    v = torch.randn(B, T, H, generator=g).bfloat16().cuda()
    lens = torch.randint(40, T + 1, (B,), generator=g)
    bias = torch.where(torch.arange(T)[None, :] < lens[:, None], 0.0, -1e30).float().cuda()   # tail padding: skipped tiles
    outs = []
    try:
        for mode in (1, 2):
            assert lib.rr_set_tuning(b"attn_fixed_ref", mode) == 0
            outs.append(_run_attn(lib, q, k, v, bias, heads))
    finally:
        lib.rr_set_tuning(b"attn_fixed_ref", -1)
    assert torch.equal(outs[0], outs[1])
    assert (outs[0] - _attn_ref(q, k, v, bias, heads)).abs().max().item() < 3e-2


@pytest.mark.parametrize("Tk", [1100, 2100])
def test_attention_long_key_sequences_reload_the_bias_chunk(lib, Tk):
    """beyond 1 024 keys the key bias / tile flags resident in LDS are reloaded every 16 tiles inside the tile loop
    (attn_block, attn_block64).  Tail-padded keys, both fixed-reference forms (32 and 64 query rows per wave) and the online
    form, enough workgroups for the fixed schedule (160 x 8 heads x 2 query blocks)."""
    B, heads, Tq = 160, 8, 200
    H = heads * 64
    g = torch.Generator().manual_seed(Tk)
    q = (torch.randn(B, Tq, H, generator=g) * 0.25).bfloat16().cuda()
    k = torch.randn(B, Tk, H, generator=g).bfloat16().cuda()
    v = torch.randn(B, Tk, H, generator=g).bfloat16().cuda()
    lens = torch.randint(Tk - 900, Tk + 1, (B,), generator=g)
    lens[0], lens[1] = Tk, 1030                                 # a full one and one that ends just inside the second chunk
    bias = torch.where(torch.arange(Tk)[None, :] < lens[:, None], 0.0, -1e30).float().cuda()
    outs = []
    try:
        for mode in (0, 1, 2):
            assert lib.rr_set_tuning(b"attn_fixed_ref", mode) == 0
            outs.append(_run_attn(lib, q, k, v, bias, heads))
    finally:
        lib.rr_set_tuning(b"attn_fixed_ref", -1)
    assert torch.equal(outs[1], outs[2])                        # the two fixed-reference forms agree bit for bit
    ref = _attn_ref(q, k, v, bias, heads)
    for o in outs:
        assert (o - ref).abs().max().item() < 3e-2


def _lo8_encode(d):
    """fp32 differences x - hi -> the e5m2 bytes of the 8-bit lo half (rr_common.h RR_LO8_SHIFT = 4), as torch rounds them."""
    return (d * 16.0).to(torch.float8_e5m2)


def _lo8_decode(b):
    return b.float() / 16.0


def _lo8_to_device_layout(b):
    """[M, N] e5m2 -> the byte buffer the residual epilogues keep (rr_common.h lo8_pair_offset): rows r and r + 16 of an aligned
    32-row group interleaved in units of 8 columns; ceil(M / 32) * 32 rows."""
    M, N = b.shape
    R = (M + 31) // 32 * 32
    u = torch.zeros(R, N, dtype=torch.uint8, device=b.device)
    u[:M] = b.view(torch.uint8)
    return u.view(R // 32, 2, 16, N // 8, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def _lo8_from_device_layout(buf, M, N):
    R = (M + 31) // 32 * 32
    return buf.view(R // 32, 16, N // 8, 2, 8).permute(0, 3, 1, 2, 4).contiguous().view(R, N)[:M].view(torch.float8_e5m2)


@pytest.mark.parametrize("lo8", [0, 1])
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("K", [128, 768])
def test_production_split_epilogue_is_bit_exact(lib, dt, K, lo8):
    """Bit-exact detector for the production epilogues gemm_kernel_hp<4, DT, 3> (VERDICT r2 weak-2 / ADVICE r2): the residual
    value the split epilogue forms internally — hi + lo, LayerNorm-recomputed with the epilogue's own expression — is
    produced as fp32 rows by rr_op_split_residual_value, the fp32-stream epilogue (itself bit-equal to the simple kernel,
    test_ln_residual_gemm_is_reproducible...) adds them, and the split epilogue's x16 / lo must equal the roundings of that
    fp32 result BIT FOR BIT (hi = operand rounding, lo = fp16(x - hi): both exact operations), four runs, in place as the
    forward does.  A lost residual term (the packed-f32 hazard) or any reordering of the epilogue arithmetic shows here.
    lo8 = 1: the same with the 8-bit lo half ("resid_lo8": e5m2 bytes of (x - hi) * 16 by the hardware's scaled pack / unpack) — the
    decoded residual value is exact in fp16, so the same fp32 reference applies, and the stored bytes must equal torch's
    round-to-nearest-even e5m2 conversion of the exact difference."""
    M, N = 512 * 256 - 31, 768
    t16 = torch.float16 if dt else torch.bfloat16
    assert lib.rr_set_op_dtype(dt) == 0 and lib.rr_set_tuning(b"resid_lo8", lo8) == 0
    try:
        g = torch.Generator().manual_seed(23 + dt + K)
        A = torch.randn(M, K, generator=g).to(t16).cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).to(t16).cuda()
        b = torch.randn(N, generator=g).cuda()
        X = (torch.randn(M, N, generator=g) * 3 + 0.5).cuda()
        hi = X.to(t16)
        lo_in8 = _lo8_encode(X - hi.float()) if lo8 else None
        lo = _lo8_decode(lo_in8).half() if lo8 else (X - hi.float()).half()      # (an e5m2 value / 16 is exact in fp16)
        if lo8:
            assert torch.equal(lo.float(), _lo8_decode(lo_in8))
        Xs = hi.float() + lo.float()
        eps = 1e-12
        st_in = torch.stack([Xs.double().mean(1), 1 / torch.sqrt(Xs.double().var(1, unbiased=False) + eps)], 1).float().contiguous()
        gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
        del X, Xs
        nparts = (N + 127) // 128
        R = torch.empty(M, N, device="cuda")
        assert lib.rr_op_split_residual_value(hi.data_ptr(), lo.data_ptr(), st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, N,
                                              R.data_ptr(), _stream()) == 0
        out32 = torch.empty(M, N, device="cuda")
        x16_f, stats_ref, part = torch.empty(M, N, device="cuda", dtype=t16), torch.empty(M, 2, device="cuda"), torch.empty(M, nparts, 2, device="cuda")
        assert lib.rr_op_gemm_resid_lnprep(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, eps, out32.data_ptr(),
                                           x16_f.data_ptr(), stats_ref.data_ptr(), part.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        want_hi = out32.to(t16)
        want_lo = _lo8_encode(out32 - want_hi.float()).view(torch.uint8) if lo8 else (out32 - want_hi.float()).half().view(torch.int16)
        assert torch.equal(want_hi, x16_f)
        del R, x16_f
        for run in range(4):
            x16, lo_out = hi.clone(), (_lo8_to_device_layout(lo_in8) if lo8 else lo.clone())   # in place: the epilogue reads and overwrites the same rows
            stats = torch.empty(M, 2, device="cuda")
            assert lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), x16.data_ptr(), lo_out.data_ptr(), st_in.data_ptr(),
                                              gamma.data_ptr(), beta.data_ptr(), M, N, K, eps, x16.data_ptr(), lo_out.data_ptr(),
                                              stats.data_ptr(), part.data_ptr(), _stream()) == 0
            torch.cuda.synchronize()
            bad = (x16.view(torch.int16) != want_hi.view(torch.int16)).nonzero()
            assert len(bad) == 0, f"run {run}: {len(bad)} hi elements differ from the fp32-stream epilogue, first {bad[:4].tolist()}"
            if lo8:
                lo_out = _lo8_from_device_layout(lo_out, M, N)
            bad = (lo_out.view(want_lo.dtype) != want_lo).nonzero()
            assert len(bad) == 0, f"run {run}: {len(bad)} lo elements differ, first {bad[:4].tolist()}"
            assert torch.allclose(stats, stats_ref, rtol=1e-6, atol=1e-6)
    finally:
        lib.rr_set_op_dtype(0)
        lib.rr_set_tuning(b"resid_lo8", -1)


@pytest.mark.parametrize("rows,cols", [(1, 128), (7, 768), (1000, 768), (33, 1024), (5, 64)])
def test_layernorm(lib, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 3 + 0.5).cuda()
    gamma = (1 + 0.1 * torch.randn(cols, generator=g)).cuda()
    beta = (0.1 * torch.randn(cols, generator=g)).cuda()
    o32 = torch.empty_like(x)
    o16 = torch.empty(rows, cols, dtype=torch.bfloat16, device="cuda")
    assert lib.rr_op_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, rows, cols, o32.data_ptr(),
                               o16.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x, (cols,), gamma, beta, 1e-12)
    assert torch.allclose(o32, ref, atol=2e-6, rtol=1e-5)
    assert torch.equal(o16, o32.bfloat16())


def test_fp16_operand_ops(lib):
    """The same GEMM / attention kernels instantiated for fp16 operands (compute_dtype "fp16")."""
    assert lib.rr_set_op_dtype(1) == 0
    try:
        g = torch.Generator().manual_seed(11)
        M, N, K = 300, 768, 768
        A = (torch.randn(M, K, generator=g) * 0.7).half().cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
        b = torch.randn(N, generator=g).cuda()
        out = torch.empty(M, N, device="cuda")
        assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 2, out.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        assert torch.allclose(out, A.float() @ W.float().t() + b, atol=3e-4, rtol=1e-4)
        outh = torch.empty(M, N, device="cuda", dtype=torch.float16)
        assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 1, outh.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        ref = _gelu(A.float() @ W.float().t() + b)
        assert ((outh.float() - ref).abs() <= 2e-3 * (1 + ref.abs())).all()
        B, heads, T = 2, 12, 200
        H = heads * 64
        q = (torch.randn(B, T, H, generator=g) * 0.5).half().cuda()
        k = torch.randn(B, T, H, generator=g).half().cuda()
        v = torch.randn(B, T, H, generator=g).half().cuda()
        keep = torch.rand(B, T, generator=g) > 0.3
        keep[:, 0] = True
        bias = torch.where(keep, 0.0, -1e30).float().cuda()
        o = torch.empty(B, T, H, dtype=torch.float16, device="cuda")
        assert lib.rr_op_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), H, H, bias.data_ptr(), B, heads, T, T,
                                        1, o.data_ptr(), H, _stream()) == 0
        torch.cuda.synchronize()
        ref = _attn_ref(q, k, v, bias, heads)
        assert (o.float() - ref).abs().max().item() < 3e-3
        x = torch.randn(9, 768, generator=g).cuda()
        gm, bt = torch.ones(768).cuda(), torch.zeros(768).cuda()
        o32, o16 = torch.empty_like(x), torch.empty(9, 768, dtype=torch.float16, device="cuda")
        assert lib.rr_op_layernorm(x.data_ptr(), gm.data_ptr(), bt.data_ptr(), 1e-12, 9, 768, o32.data_ptr(),
                                   o16.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(o16, o32.half())
    finally:
        lib.rr_set_op_dtype(0)


# ---- every tile configuration behind rr_set_gemm_variant, the production half-tile-ring kernel (11 direct / 12 LDS
# epilogue) included: the shape heuristic only picks it at >= 512 output tiles, so it is forced here on ragged shapes
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 10, 11, 12, 14])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4, 5])
def test_gemm_every_variant_every_epilogue(lib, variant, epi):
    shapes = [(1000, 768, 768), (515, 2304, 768), (257, 200, 3072), (64, 64, 64)]
    try:
        if lib.rr_set_gemm_variant(variant) != 0:
            pytest.skip(f"variant {variant} not built")
        for (M, N, K) in shapes:
            g = torch.Generator(device="cpu").manual_seed(M + N + K + 13 * epi)
            A = (torch.randn(M, K, generator=g) * 0.7).bfloat16().cuda()
            W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
            b = torch.randn(N, generator=g).cuda()
            R = torch.randn(M, N, generator=g).cuda()
            f32_out = epi in (2, 4)
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if f32_out else torch.bfloat16)
            if epi == 4:
                rc = lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K,
                                              out.data_ptr(), _stream())
            else:
                rc = lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), _stream())
            assert rc == 0, (variant, epi, M, N, K)
            torch.cuda.synchronize()
            ref = A.float() @ W.float().t() + b
            ref = {0: ref, 1: _gelu(ref), 2: ref, 3: torch.tanh(ref), 4: ref + R, 5: ref * torch.sigmoid(1.702 * ref)}[epi]
            got = out.float()
            assert torch.isfinite(got).all(), (variant, epi, M, N, K)
            tol = 2e-4 if f32_out else 1.2e-2
            err = (got - ref).abs()
            assert (err <= tol * (1 + ref.abs())).all(), f"variant {variant} epi {epi} {M}x{N}x{K}: max err {err.max().item()}"
    finally:
        lib.rr_set_gemm_variant(-1)


@pytest.mark.parametrize("dt", [0, 1])
def test_gemm_rows_do_not_depend_on_where_their_tile_lies(lib, dt):
    """The K walk of every second block of 1 024 output columns runs backwards (gemm_bf16.hip k_walk_reversed: an XCD's next round
    of the persistent ring starts on the K slices its L2 still holds).  The direction is a function of the COLUMN alone, in every
    16-bit GEMM kernel, so a row's values depend neither on M, nor on the kernel the shape heuristic picks, nor on the row's place
    in the matrix — what packed == bucketed == padded rests on.  Ring kernel (66 000 rows) against the simple kernel on 300 of
    those rows taken from the middle, FFN-up and QKV widths (12 and 9 column slices), fp32 output: bit for bit; and the reversed
    columns are as close to the fp64 product as the forward ones."""
    t16 = torch.float16 if dt else torch.bfloat16
    assert lib.rr_set_op_dtype(dt) == 0
    try:
        for (N, K) in [(3072, 768), (2304, 768), (4096, 1024)]:
            M, r0, m = 66_000, 31_111, 300
            g = torch.Generator(device="cpu").manual_seed(N + K + dt)
            A = torch.randn(M, K, generator=g).to(t16).cuda()
            W = (torch.randn(N, K, generator=g) * 0.05).to(t16).cuda()
            b = torch.randn(N, generator=g).cuda()
            big = torch.empty(M, N, device="cuda")
            assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 2, big.data_ptr(), _stream()) == 0
            sub_in = A[r0:r0 + m].contiguous()
            small = torch.empty(m, N, device="cuda")
            assert lib.rr_op_gemm_bf16(sub_in.data_ptr(), W.data_ptr(), b.data_ptr(), m, N, K, 2, small.data_ptr(), _stream()) == 0
            torch.cuda.synchronize()
            assert torch.equal(big[r0:r0 + m], small), (N, K, (big[r0:r0 + m] - small).abs().max().item())
            ref = sub_in.double() @ W.double().t() + b.double()
            err = (small.double() - ref).abs().view(m, N // 256, 256).amax(dim=(0, 2))
            fwd = torch.tensor([((c >> 2) & 1) == 0 for c in range(N // 256)], device="cuda")
            assert err.max().item() < 2e-5 and err[~fwd].max().item() < 1.5 * err[fwd].max().item() + 1e-6, err.tolist()
    finally:
        lib.rr_set_op_dtype(0)


@pytest.mark.parametrize("dt", [0, 1])
def test_small_grid_half_row_tiles_are_bit_identical(lib, dt):
    """Below two 128 x 128 workgroups per CU the two-stage kernel runs 64 x 128 tiles (gemm_bf16.hip half_rows, the strong-scaling
    shard shapes): tile shape moves time only — every epilogue of the shard step's small GEMMs, with the switch off and on
    (rr_set_tuning "gemm_small_half_rows"), bit for bit, incl. ragged last tiles."""
    t16 = torch.float16 if dt else torch.bfloat16
    assert lib.rr_set_op_dtype(dt) == 0
    try:
        for (M, N, K) in [(6656, 768, 768), (6656, 768, 3072), (777, 1024, 256), (13, 768, 768)]:
            g = torch.Generator(device="cpu").manual_seed(M + N + K + dt)
            A = (torch.randn(M, K, generator=g) * 0.7).to(t16).cuda()
            W = (torch.randn(N, K, generator=g) * 0.05).to(t16).cuda()
            b = torch.randn(N, generator=g).cuda()
            R = torch.randn(M, N, generator=g).cuda()
            outs = {}
            for on in (0, 1):
                assert lib.rr_set_tuning(b"gemm_small_half_rows", on) == 0
                res = []
                for epi in (0, 1, 2):
                    o = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if epi == 2 else t16)
                    assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, o.data_ptr(), _stream()) == 0
                    res.append(o)
                o = torch.full((M, N), float("nan"), device="cuda")
                assert lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, o.data_ptr(), _stream()) == 0
                res.append(o)
                torch.cuda.synchronize()
                outs[on] = res
            for a, c in zip(outs[0], outs[1]):
                assert torch.isfinite(a.float()).all() and torch.equal(a, c), (M, N, K)
    finally:
        lib.rr_set_tuning(b"gemm_small_half_rows", 1)
        lib.rr_set_op_dtype(0)


def test_gemm_production_kernel_equals_simple_kernel_at_bench_shape(lib):
    """The four bert-base GEMM shapes at a bench-sized M (heuristic -> half-tile-ring kernel) against the simple
    128x128 loop on the same operands: same products and fp32 accumulation, only the summation order differs."""
    M = 256 * 180 + 77
    for (N, K, epi) in [(2304, 768, 0), (768, 768, 4), (3072, 768, 1), (768, 3072, 4)]:
        g = torch.Generator(device="cpu").manual_seed(N + K)
        A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
        b = torch.randn(N, generator=g).cuda()
        R = torch.randn(M, N, generator=g).cuda()
        outs = []
        for variant in (-1, 0):
            lib.rr_set_gemm_variant(variant)
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if epi == 4 else torch.bfloat16)
            if epi == 4:
                rc = lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K,
                                              out.data_ptr(), _stream())
            else:
                rc = lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, epi, out.data_ptr(), _stream())
            assert rc == 0
            torch.cuda.synchronize()
            outs.append(out.float())
        lib.rr_set_gemm_variant(-1)
        d = (outs[0] - outs[1]).abs()
        assert torch.isfinite(outs[0]).all()
        if epi == 4:
            assert d.max().item() <= 2e-4
        else:   # 16-bit outputs: at most one ulp apart where the fp32 sums straddle a rounding boundary
            assert (d <= 2.0 ** -7 * (outs[1].abs() + 1e-3)).all() and (d > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("variant", [0, 1, 2, 10, 11, 12, 14])
def test_ln_residual_gemm_is_reproducible_and_matches_materialised_residual(lib, variant):
    """out = A W^T + b + LN(x) with the residual recomputed from (x, mean, rstd, gamma, beta) inside the epilogue must be
    bit-identical to the same GEMM fed the fp32 LayerNorm output, and identical run to run, at a size where every CU
    holds several workgroups (this caught lost residual terms from compiler-packed f32 code; see build.py)."""
    M, N, K = 200 * 512, 768, 3072
    g = torch.Generator(device="cpu").manual_seed(11)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    gam = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    bet = (0.05 * torch.randn(N, generator=g)).cuda()
    ln32 = torch.empty(M, N, device="cuda")
    ln16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    stats = torch.empty(M, 2, device="cuda")
    assert lib.rr_op_layernorm_stats(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-12, M, N, ln32.data_ptr(),
                                     ln16.data_ptr(), stats.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    ref_stats = torch.stack([x.mean(1), torch.rsqrt(x.var(1, unbiased=False) + 1e-12)], 1)
    assert torch.allclose(stats, ref_stats, atol=1e-5, rtol=1e-5)
    try:
        assert lib.rr_set_gemm_variant(variant) == 0
        plain = torch.empty(M, N, device="cuda")
        assert lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), ln32.data_ptr(), M, N, K,
                                        plain.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        for _ in range(4):
            out = torch.full((M, N), float("nan"), device="cuda")
            assert lib.rr_op_gemm_ln_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(), stats.data_ptr(),
                                               gam.data_ptr(), bet.data_ptr(), M, N, K, out.data_ptr(), _stream()) == 0
            torch.cuda.synchronize()
            bad = (out != plain).nonzero()
            assert len(bad) == 0, f"variant {variant}: {len(bad)} elements differ, first {bad[:4].tolist()}"
    finally:
        lib.rr_set_gemm_variant(-1)


@pytest.mark.parametrize("variant", [12, 14])
@pytest.mark.parametrize("M,N,K", [(70001, 1000, 128), (33 * 256 + 5, 2304, 64), (256 * 300, 768, 192), (65537, 264, 320)])
def test_ring_kernels_ragged_multi_tile(lib, variant, M, N, K):
    """The ring kernels at sizes where a persistent workgroup walks several tiles, with ragged last row/column tiles and
    one- to five-tile K loops (steady-state loop, guarded tail and the K-tile-0-only case), against fp32 torch."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda()
    ref = A.float() @ W.float().t() + b
    try:
        assert lib.rr_set_gemm_variant(variant) == 0
        out16 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 0, out16.data_ptr(), _stream()) == 0
        out32 = torch.full((M, N), float("nan"), device="cuda")
        assert lib.rr_op_gemm_resid_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, out32.data_ptr(),
                                        _stream()) == 0
        torch.cuda.synchronize()
    finally:
        lib.rr_set_gemm_variant(-1)
    e16 = (out16.float() - ref).abs()
    assert torch.isfinite(out16.float()).all() and (e16 <= 1.2e-2 * (1 + ref.abs())).all(), e16.max().item()
    e32 = (out32 - (ref + R)).abs()
    assert torch.isfinite(out32).all() and (e32 <= 2e-4 * (1 + ref.abs())).all(), e32.max().item()


@pytest.mark.parametrize("variant", [-1, 0])     # -1: the production heuristic (persistent ring), 0: the 128x128 kernel
def test_gemm_operand_beyond_4_gib(lib, variant):
    """A of 4.3 GB (700 000 rows x 3072, the FFN-down operand of ~1 370 pairs at S = 512): row addresses are a 64-bit
    scalar tile origin plus a 32-bit in-tile offset, so nothing wraps.  Checked on row blocks before, across and after the
    4 GiB boundary (row 699 050) and on the last, partial tile."""
    M, N, K = 700_000, 768, 3072
    assert M * K * 2 > 1 << 32
    A = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for r0 in range(0, M, 100_000):                                       # fill in chunks (keeps the fp32 temporary small)
        A[r0: r0 + 100_000] = torch.randn(min(100_000, M - r0), K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    try:
        assert lib.rr_set_gemm_variant(variant) == 0
        assert lib.rr_op_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, 0, out.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
    finally:
        lib.rr_set_gemm_variant(-1)
    edge = (1 << 32) // (K * 2)
    for r0 in (0, 350_000, edge - 300, edge + 300, M - 300):
        rows = slice(r0, min(M, r0 + 300))
        ref = A[rows].float() @ W.float().t() + b
        err = (out[rows].float() - ref).abs()
        assert torch.isfinite(out[rows].float()).all()
        assert (err <= 1.2e-2 * (1 + ref.abs())).all(), f"rows {r0}: max err {err.max().item()}"


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (2000, 768, 3072), (70_000, 768, 768), (1500, 1024, 1024), (77, 128, 256)])
def test_folded_layernorm_halves(lib, M, N, K):
    """Producer: residual GEMM that also emits the 16-bit rows and (mean, rstd); consumer: GEMM on the raw rows with the
    LayerNorm applied to the accumulators.  Small shapes run the 128x128 kernel, 70 000 x 768 the persistent ring."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = (torch.randn(M, N, generator=g) * 2 + 0.7).cuda()                     # rows with a mean
    out = torch.empty(M, N, device="cuda")
    x16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    stats = torch.empty(M, 2, device="cuda")
    part = torch.empty(M, (N + 127) // 128, 2, device="cuda")
    eps = 1e-12
    assert lib.rr_op_gemm_resid_lnprep(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, eps, out.data_ptr(),
                                       x16.data_ptr(), stats.data_ptr(), part.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + b + R
    assert torch.allclose(out, ref, atol=5e-4, rtol=1e-4)
    assert torch.equal(x16, out.bfloat16())                                   # the very same values, rounded once
    mean, var = out.double().mean(1), out.double().var(1, unbiased=False)
    assert torch.allclose(stats[:, 0].double(), mean, atol=2e-6, rtol=1e-6)
    assert torch.allclose(stats[:, 1].double(), 1 / torch.sqrt(var + eps), rtol=2e-6)
    # consumer: LayerNorm(out) W2^T + b2 from the raw 16-bit rows
    N2 = 256
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
    W2 = (torch.randn(N2, N, generator=g) * 0.03).cuda()
    b2 = torch.randn(N2, generator=g).cuda()
    Wf = (W2 * gamma[None, :]).bfloat16()
    csum = Wf.double().sum(1).float()
    dvec = (W2.double() @ beta.double()).float() + b2
    for epi, dtype in ((0, torch.bfloat16), (1, torch.bfloat16), (2, torch.float32)):
        o2 = torch.empty(M, N2, device="cuda", dtype=dtype)
        assert lib.rr_op_gemm_lnfold(x16.data_ptr(), Wf.data_ptr(), dvec.data_ptr(), csum.data_ptr(), stats.data_ptr(), M, N2, N,
                                     epi, o2.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        # exact arithmetic on the same operands: rstd * (x16 Wf^T - mean * csum) + d
        want = stats[:, 1:2] * (x16.float() @ Wf.float().t() - stats[:, 0:1] * csum[None, :]) + dvec
        if epi == 1:
            want = _gelu(want)
        tol = 3e-4 if epi == 2 else 1.2e-2
        err = (o2.float() - want).abs()
        assert (err <= tol * (1 + want.abs())).all(), f"epi {epi}: max err {err.max().item()}"
        # and it IS the LayerNorm-then-Linear of the fp32 rows, up to the 16-bit operand rounding
        ln = torch.nn.functional.layer_norm(out, (N,), gamma, beta, eps) @ W2.t() + b2
        if epi == 1:
            ln = _gelu(ln)
        assert (o2.float() - ln).abs().max().item() < 0.06


@pytest.mark.parametrize("dt", [0, 1])
def test_eight_bit_lo_half_at_extreme_magnitudes(lib, dt):
    """The 8-bit lo half where its e5m2 code leaves the comfortable range: residual columns of magnitude 1e3 and 1e-4 beside
    ordinary ones, output columns of magnitude 2e4 (lo * 16 up to 256) and 1e-5 (lo * 16 below the smallest e5m2 subnormal:
    the code underflows to zero, i.e. the element keeps hi's precision, an absolute error below 2^-21).  Same bit-exact
    construction as test_production_split_epilogue_is_bit_exact: the bytes must be torch's round-to-nearest-even e5m2
    conversion of the exact difference — subnormals included — and the decoded rows must reproduce the fp32 rows to three
    significant bits of lo or 2^-21 absolute."""
    M, N, K = 512 * 256 - 31, 768, 128
    t16 = torch.float16 if dt else torch.bfloat16
    assert lib.rr_set_op_dtype(dt) == 0 and lib.rr_set_tuning(b"resid_lo8", 1) == 0
    try:
        g = torch.Generator().manual_seed(91 + dt)
        A = torch.randn(M, K, generator=g).to(t16).cuda()
        Wf = torch.randn(N, K, generator=g) * 0.05
        Wf[128:192] = 0.0                                           # columns 128..191: no GEMM term
        W = Wf.to(t16).cuda()
        b = torch.randn(N, generator=g)
        b[128:192] *= 1e-5                                           # ... and a bias of 1e-5: outputs the e5m2 code of lo cannot reach
        b[192:256] = 2.0e4 * (1.0 + 0.01 * torch.randn(64, generator=g))        # outputs near 2e4: lo * 16 up to 256
        b = b.cuda()
        X = torch.randn(M, N, generator=g) * 3 + 0.5
        X[:, 0:64] *= 1.0e3
        X[:, 64:128] *= 1.0e-4
        X = X.cuda()
        hi = X.to(t16)
        lo_in8 = _lo8_encode(X - hi.float())
        lo = _lo8_decode(lo_in8).half()
        assert torch.equal(lo.float(), _lo8_decode(lo_in8))          # an e5m2 value / 16 is exact in fp16, subnormals included
        Xs = hi.float() + lo.float()
        assert ((Xs - X).abs() <= X.abs() * 2.0 ** (-11 if dt == 0 else -14) + 2.0 ** -21).all()
        eps = 1e-12
        st_in = torch.stack([Xs.double().mean(1), 1 / torch.sqrt(Xs.double().var(1, unbiased=False) + eps)], 1).float().contiguous()
        gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
        gamma[128:192] = 0.0
        beta[128:192] = 0.0
        del X, Xs
        nparts = (N + 127) // 128
        R = torch.empty(M, N, device="cuda")
        assert lib.rr_op_split_residual_value(hi.data_ptr(), lo.data_ptr(), st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, N,
                                              R.data_ptr(), _stream()) == 0
        out32 = torch.empty(M, N, device="cuda")
        x16_f, stats_ref, part = torch.empty(M, N, device="cuda", dtype=t16), torch.empty(M, 2, device="cuda"), torch.empty(M, nparts, 2, device="cuda")
        assert lib.rr_op_gemm_resid_lnprep(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), M, N, K, eps, out32.data_ptr(),
                                           x16_f.data_ptr(), stats_ref.data_ptr(), part.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        assert out32[:, 128:192].abs().max().item() < 1e-4 and out32[:, 192:256].abs().min().item() > 1.5e4
        want_hi = out32.to(t16)
        want_lo = _lo8_encode(out32 - want_hi.float()).view(torch.uint8)
        del R, x16_f
        x16, lo_out = hi.clone(), _lo8_to_device_layout(lo_in8)
        stats = torch.empty(M, 2, device="cuda")
        assert lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), x16.data_ptr(), lo_out.data_ptr(), st_in.data_ptr(),
                                          gamma.data_ptr(), beta.data_ptr(), M, N, K, eps, x16.data_ptr(), lo_out.data_ptr(),
                                          stats.data_ptr(), part.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(x16.view(torch.int16), want_hi.view(torch.int16))
        got_lo = _lo8_from_device_layout(lo_out, M, N).view(torch.uint8)
        bad = (got_lo != want_lo).nonzero()
        assert len(bad) == 0, f"{len(bad)} lo bytes differ from torch's e5m2 rounding, first {bad[:4].tolist()}: " \
                              f"{[(int(got_lo[i, j]), int(want_lo[i, j]), float(out32[i, j])) for i, j in bad[:4].tolist()]}"
        got = x16.float() + _lo8_decode(got_lo.view(torch.float8_e5m2))
        assert ((got - out32).abs() <= out32.abs() * 2.0 ** (-11 if dt == 0 else -14) + 2.0 ** -21).all()
    finally:
        lib.rr_set_op_dtype(0)
        lib.rr_set_tuning(b"resid_lo8", -1)


@pytest.mark.parametrize("lo8", [0, 1])
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("with_ln,in_place", [(False, False), (True, True)])
def test_split_residual_stream_epilogue(lib, dt, with_ln, in_place, lo8):
    """The residual epilogue on the split stream (rr_op_gemm_resid_split): residual rows as hi (operand type) + lo (fp16),
    optionally LayerNormed on the fly, output rows as (x16, lo) + statistics — against the fp32-stream epilogue
    (rr_op_gemm_resid_lnprep) fed the SAME residual values.  x16 must be bit-identical, hi + lo must reproduce the fp32
    rows to 2^-20 relative (fp16 operands: 2^-22), the statistics must agree, and the in-place form (what the forward
    does) must equal the out-of-place one.  130 048 x 768 = 1 524 tiles: the persistent ring kernel; ragged last row tile.
    lo8 = 1 ("resid_lo8"): lo as e5m2 bytes of (x - hi) * 16 — hi + lo then reproduces the fp32 rows to 2^-14 (fp16 hi) / 2^-11
    (bf16 hi) of the element's binade: three significant bits of a lo that is at most half an ulp of hi."""
    M, N, K = 130_048 - 77, 768, 128
    t16 = torch.float16 if dt else torch.bfloat16
    assert lib.rr_set_op_dtype(dt) == 0 and lib.rr_set_tuning(b"resid_lo8", lo8) == 0
    try:
        g = torch.Generator().manual_seed(17 + dt)
        A = torch.randn(M, K, generator=g).to(t16).cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).to(t16).cuda()
        b = torch.randn(N, generator=g).cuda()
        X = (torch.randn(M, N, generator=g) * 3 + 0.5).cuda()                   # previous sublayer's pre-LayerNorm rows
        hi = X.to(t16)
        lo = _lo8_encode(X - hi.float()) if lo8 else (X - hi.float()).half()
        Xs = hi.float() + (_lo8_decode(lo) if lo8 else lo.float())               # what the split stream carries
        rel = 2.0 ** ((-11 if dt == 0 else -14) if lo8 else (-19 if dt == 0 else -21))
        assert ((Xs - X).abs() <= X.abs() * rel + (5e-7 if lo8 else 1e-7)).all()
        eps = 1e-12
        nparts = (N + 127) // 128
        st_in = gamma = beta = None
        R = Xs
        if with_ln:
            mu, var = Xs.double().mean(1), Xs.double().var(1, unbiased=False)
            st_in = torch.stack([mu, 1 / torch.sqrt(var + eps)], 1).float().contiguous()
            gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.05 * torch.randn(N, generator=g)).cuda()
            R = ((Xs - st_in[:, 0:1]) * st_in[:, 1:2] * gamma + beta)           # the kernel's own expression, fp32
        # fp32-stream epilogue on the same residual values
        out32 = torch.empty(M, N, device="cuda")
        x16_ref = torch.empty(M, N, device="cuda", dtype=t16)
        stats_ref, part = torch.empty(M, 2, device="cuda"), torch.empty(M, nparts, 2, device="cuda")
        assert lib.rr_op_gemm_resid_lnprep(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.contiguous().data_ptr(), M, N, K, eps,
                                           out32.data_ptr(), x16_ref.data_ptr(), stats_ref.data_ptr(), part.data_ptr(), _stream()) == 0
        # split-stream epilogue
        lo_dev = _lo8_to_device_layout(lo) if lo8 else lo
        if in_place:
            x16, lo_out = hi.clone(), lo_dev.clone()
            hi_in, lo_in = x16, lo_out
        else:
            x16, lo_out = torch.empty_like(hi), torch.empty_like(lo_dev)
            hi_in, lo_in = hi, lo_dev
        stats = torch.empty(M, 2, device="cuda")
        P = lambda t: t.data_ptr() if t is not None else 0
        assert lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), hi_in.data_ptr(), lo_in.data_ptr(), P(st_in),
                                          P(gamma), P(beta), M, N, K, eps, x16.data_ptr(), lo_out.data_ptr(), stats.data_ptr(),
                                          part.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        if with_ln:     # the reference residual was normalised by torch (no fused multiply-add): 1-ulp differences in fp32 flip a few roundings
            dx = (x16.float() - x16_ref.float()).abs()
            assert (dx <= x16_ref.float().abs() * 2.0 ** (-7 if dt == 0 else -10) + 4e-6).all() and (dx > 0).float().mean().item() < 2e-3
        else:
            assert torch.equal(x16, x16_ref)
        got = x16.float() + (_lo8_decode(_lo8_from_device_layout(lo_out, M, N)) if lo8 else lo_out.float())
        assert ((got - out32).abs() <= out32.abs() * rel + (2e-6 if with_ln else 2e-7) + (5e-7 if lo8 else 0.0)).all(), (got - out32).abs().max().item()
        assert torch.allclose(stats, stats_ref, rtol=1e-6, atol=1e-6)
        # shapes the ring kernel does not run are refused, not silently computed some other way
        assert lib.rr_op_gemm_resid_split(A.data_ptr(), W.data_ptr(), b.data_ptr(), hi.data_ptr(), lo_dev.data_ptr(), 0, 0, 0, 512, N, K,
                                          eps, x16.data_ptr(), lo_out.data_ptr(), stats.data_ptr(), part.data_ptr(), _stream()) == -4
    finally:
        lib.rr_set_op_dtype(0)
        lib.rr_set_tuning(b"resid_lo8", -1)
