"""GPU parity at BASELINE.json's FULL sizes against committed stock-HF goldens (tests/golden/make_golden.py
run_fullsize_case): c3_full (K = 100, S = 512, 81 vision tokens), c5_full (bert-large, K = 200, S = 512), l_shape
(monoPreFLMR-L geometry: 1024-d ViT-L/14 features, 288 vision tokens, T = 800, position table 900) and c3_sep, the
Recall@5 fixture whose fp32 logits leave a designed gap between rank 5 and rank 6.

Gates: compute_dtype = fp16 (the headline mode): |logit - fp32| <= 1e-3 (north_star) on every golden, the 25-layer bert-large
stack included (measured 8.6e-4 there; the reference's own bf16-mixed forward is 1e-2 from fp32); compute_dtype = bf16:
helpers.bf16_gate (the reference's own autocast drift on the same inputs).  Every measured margin is written to
gpurun_out/parity_margins.json by helpers.record_margin and folded into profiles/rNN_parity_margins.json by
tools/merge_margins.py (which never drops a key)."""
import numpy as np
import pytest
import torch

from helpers import O, arch_from_cfg, load_fullsize, margin_stats, record_margin

pytestmark = pytest.mark.gpu


def _engine(cfg, vision, w, dt):
    import rmr_amd
    eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, vision, dt))
    eng.load_state_dict(w)
    return eng


def _logits(eng, q, sel=None, want_order=False):
    ids, am, tt = (x if sel is None else x[sel] for x in (q["ids"], q["am"], q["tt"]))
    K = ids.shape[0]
    img = q["img"]
    r = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), 1, K, None if img[0] is None else img[0].cuda(),
                        None if img[1] is None else img[1].cuda(), None, want_order=want_order)
    torch.cuda.synchronize()
    return r


# measured at HEAD, one -m gpu run (the figures are checked against the record by tests/test_docs_cpu.py): fp16
# 3.4e-4 [`profiles/r05_parity_margins.json` "c3_full/fp16" "max_abs"], 2.6e-4 [`profiles/r05_parity_margins.json` "l_shape/fp16" "max_abs"],
# 9.3e-4 [`profiles/r05_parity_margins.json` "c5_full/fp16" "max_abs"] — north_star's 1e-3 on all three, the 25-layer bert-large stack included
# (the round-2 gate there was 2e-3); bf16 2.8e-3 [`profiles/r05_parity_margins.json` "c3_full/bf16" "max_abs"],
# 1.7e-3 [`profiles/r05_parity_margins.json` "l_shape/bf16" "max_abs"], 3.4e-3 [`profiles/r05_parity_margins.json` "c5_full/bf16" "max_abs"] against its own gate
@pytest.mark.parametrize("name,tol16", [("c3_full", 1e-3), ("l_shape", 1e-3), ("c5_full", 1e-3)])
def test_full_size_logits_match_the_fp32_goldens(name, tol16):
    cfg, w, vision, qs = load_fullsize(name)
    q = qs[0]
    ac = (q["autocast"] - q["fp32"]).abs().max().item()
    for dt in ("fp16", "bf16"):
        eng = _engine(cfg, vision, w, dt)
        lg = _logits(eng, q)["logits"].cpu()
        d = (lg - q["fp32"]).abs().max().item()
        gate = tol16 if dt == "fp16" else max(1e-3, 1.5 * ac)
        print(f"[{name}/{dt}] K={len(lg)} |dlogit| vs fp32 {d:.2e} (gate {gate:.1e}); reference's bf16-autocast drift {ac:.2e}")
        record_margin(f"{name}/{dt}", gate=gate, reference_bf16_autocast_drift=ac, **margin_stats(lg, q["fp32"]))
        assert torch.isfinite(lg).all()
        assert d <= gate
        del eng
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_recall_at_5_and_top5_sets_match_the_fp32_reference(dt):
    """c3_sep: two queries x 100 candidates at S = 512 with vision; the fp32 stock-HF logits leave >= 0.08 between rank 5
    and rank 6 (the bf16 rounding noise of this widened random network is ~2e-2), query 0's only positive is the fp32
    rank-5 candidate and query 1's the rank-6 one.  Asserted unconditionally: identical top-5 id SETS, Recall@5 = (1, 0)
    exactly as the fp32 reference ranks them, Recall@10 = 1 for both, and the device order is the stable descending sort
    of the device logits."""
    import rmr_amd
    from rmr_amd import _lib
    cfg, w, vision, qs = load_fullsize("c3_sep")
    eng = _engine(cfg, vision, w, dt)
    ranked, pos, ranked_ref = [], [], []
    # c3_sep's widened weights (gain 2.5) give PEAKED attention: the case in which the fixed-reference schedule (100 pairs x 12
    # heads x 4-5 query blocks >= 1 024 workgroups: it is the one that runs here) may have to recompute workgroups online
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.load().rr_set_attn_redo_stats(cnt.data_ptr())
    for qi, q in enumerate(qs):
        sel = torch.from_numpy(q["selected"].astype(np.int64))
        assert float(q["gap_5_6"]) >= 0.08
        r = _logits(eng, q, sel, want_order=True)
        lg = r["logits"].cpu()
        ref = q["fp32"][sel]
        order = r["order"][0].cpu().tolist()
        ref_order = O.rank_descending_stable(ref.tolist())
        d = (lg - ref).abs().max().item()
        print(f"[c3_sep/{dt} q{qi}] |dlogit| vs fp32 {d:.2e}, logit std {ref.std():.3f}, rank-5/6 gap {float(q['gap_5_6']):.3f}")
        record_margin(f"c3_sep/q{qi}/{dt}", gate=0.5 * float(q["gap_5_6"]), gap_5_6=float(q["gap_5_6"]), **margin_stats(lg, ref))
        assert order == O.rank_descending_stable(lg.tolist())
        assert set(order[:5]) == set(ref_order[:5])
        assert d < 0.5 * float(q["gap_5_6"])
        rho = torch.corrcoef(torch.stack([torch.tensor(ref_order).argsort().float(), torch.tensor(order).argsort().float()]))[0, 1]
        assert rho > 0.98
        ranked.append(order)
        ranked_ref.append(ref_order)
        pos.append([int(q["positive_list_index"])])
    torch.cuda.synchronize()
    _lib.load().rr_set_attn_redo_stats(0)
    record_margin(f"c3_sep/{dt}/attention_redo", workgroups_recomputed_online=int(cnt[0]), workgroups=int(cnt[1]),
                  fraction=float(cnt[0]) / max(1, int(cnt[1])))
    assert int(cnt[1]) > 0                                     # the fixed-reference schedule ran
    got = rmr_amd.recall_precision_at_k(ranked, pos, [5, 10])
    want = O.recall_precision_at_k(ranked_ref, pos, [5, 10])
    assert got == want
    assert got["recall"] == [0.5, 1.0]                       # query 0 hits at rank 5, query 1 only at rank 6


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_split_residual_stream_costs_no_accuracy(dt):
    """VERDICT r2 item 2c.  The split residual stream (hi = the 16-bit operand rows, lo = fp16 remainder) carries 22 (fp16
    operands) / 19 (bf16) bits of every pre-LayerNorm row instead of 24.  Its logits differ from the fp32-stream build's by
    RE-DECIDED operand roundings downstream (a 2^-19 relative perturbation of a row flips the 8-bit rounding of ~1 element
    in 2 000, each flip moves an operand by 2^-8): in bf16 mode that is the same chaotic ~2e-3 any two correct bf16
    forwards differ by (DESIGN.md "Numerics"), not lost accuracy.  What is gated is therefore the distance to the fp32
    GOLDEN: with the split stream it must not be worse than with fp32 rows beyond 15 % (+ 1e-4), and in fp16 mode the two
    builds must agree within 5e-4.  Both distances and the build-to-build difference are recorded."""
    from rmr_amd import _lib
    lib = _lib.load()
    cfg, w, vision, qs = load_fullsize("c3_full")
    q = qs[0]
    eng = _engine(cfg, vision, w, dt)
    try:
        on = _logits(eng, q)["logits"].cpu()
        assert lib.rr_set_tuning(b"resid_split", 0) == 0
        off = _logits(eng, q)["logits"].cpu()
    finally:
        lib.rr_set_tuning(b"resid_split", 1)
    d_on, d_off = (on - q["fp32"]).abs().max().item(), (off - q["fp32"]).abs().max().item()
    between = (on - off).abs().max().item()
    print(f"[c3_full/{dt}] vs fp32 golden: split stream {d_on:.2e}, fp32 stream {d_off:.2e}; split vs fp32 stream {between:.2e}")
    record_margin(f"c3_full/{dt}/split_vs_fp32_stream", split_vs_golden=d_on, fp32_stream_vs_golden=d_off, split_vs_fp32_stream=between)
    assert d_on <= 1.15 * d_off + 1e-4
    if dt == "fp16":
        assert between <= 5e-4


@pytest.mark.parametrize("name", ["c3_full", "c5_full"])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_eight_bit_lo_half_costs_no_accuracy(name, dt):
    """Handle option "resid_lo8": the lo half of the split residual stream as e5m2 bytes of (x - hi) * 16 — 3 significant bits of
    a remainder that is at most half an ulp of hi, i.e. x to 14 bits behind an fp16 hi (the GEMMs read 11 of them), 11 behind a
    bf16 hi — instead of fp16 (22 / 19 bits).  Two correct forwards of this network differ by re-decided operand roundings at
    the 2e-4 (fp16) / 2e-3 (bf16) level whatever the size of the perturbation (test_split_residual_stream_costs_no_accuracy), so
    what is gated is the distance to the fp32 GOLDEN: the 8-bit form must meet the mode's own gate (north_star's 1e-3 in fp16,
    1.5 x the reference's autocast drift in bf16) and, in fp16 — where it is the default —, must not be worse than the fp16 lo
    beyond 15 % (+ 1e-4: the yardstick of the split-stream test above).  Also: the default follows the operand type."""
    cfg, w, vision, qs = load_fullsize(name)
    q = qs[0]
    ac = (q["autocast"] - q["fp32"]).abs().max().item()
    eng = _engine(cfg, vision, w, dt)
    assert eng.get_option("resid_lo8") == (1 if dt == "fp16" else 0)
    d = {}
    for lo8 in (1, 0):
        eng.set_option("resid_lo8", lo8)
        assert eng.get_option("resid_lo8") == lo8
        lg = _logits(eng, q)["logits"].cpu()
        assert torch.isfinite(lg).all()
        d[lo8] = (lg - q["fp32"]).abs().max().item()
        d[(lo8, "logits")] = lg
    between = (d[(1, "logits")] - d[(0, "logits")]).abs().max().item()
    gate = 1e-3 if dt == "fp16" else max(1e-3, 1.5 * ac)
    print(f"[{name}/{dt}] vs fp32 golden: 8-bit lo {d[1]:.2e}, fp16 lo {d[0]:.2e} (gate {gate:.1e}); between the two {between:.2e}")
    record_margin(f"{name}/{dt}/lo8_vs_lo16", lo8_vs_golden=d[1], lo16_vs_golden=d[0], lo8_vs_lo16=between, gate=gate)
    assert d[1] <= gate and d[0] <= gate
    if dt == "fp16":
        assert d[1] <= 1.15 * d[0] + 1e-4


@pytest.mark.parametrize("vision,dt", [(False, "fp16"), (True, "fp16"), (False, "bf16")])
def test_packed_forward_at_the_bench_size(vision, dt):
    """rr_forward_packed at the headline size (8 queries x 100 candidates, S = 512, pair lengths U[64, 512], the c3 shape with
    and without its 81 vision tokens): eight length groups, 258 k of 410 k rows.  Size-independent properties: every pair's
    logit equals the padded forward's (bit for bit text-only; fp32 summation order of the cross-encoder's attention with
    vision tokens), the rank order of every list is the padded one where the logits allow, and a second call reproduces
    the first bit for bit."""
    import rmr_amd
    from rmr_amd.synthetic import image_features, pair_batch
    arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1, cross_encoder_max_position_embeddings=750, loss_fn="BCE"),
                             has_vision=int(vision), compute_dtype=dt)
    eng = rmr_amd.RerankEngine(arch)
    eng.load_state_dict(rmr_amd.synthetic_state_dict(arch, 0, True))
    Bq, K, S = 8, 100, 512
    ids, am, tt = [t.cuda() for t in pair_batch(arch["vocab_size"], Bq, K, S, regime="realistic")]
    cls, pat = [t.cuda() for t in image_features(Bq, arch["n_patches"], arch["vision_hidden"])] if vision else (None, None)
    ref = eng.forward_ids(ids, am, tt, Bq, K, cls, pat, want_order=True)
    got = eng.forward_ids_packed(ids, am, tt, Bq, K, cls, pat, granule=64, want_order=True)
    again = eng.forward_ids_packed(ids, am, tt, Bq, K, cls, pat, granule=64, want_order=True,
                                   lengths=((ids != 0) | (am != 0)).long().mul(torch.arange(1, S + 1, device="cuda")).amax(1).cpu().tolist())
    torch.cuda.synchronize()
    assert got["packed_rows"] < 0.7 * Bq * K * S
    assert torch.equal(got["logits"], again["logits"])
    assert not eng.activation_range_exceeded()           # the fp16 range guard stays quiet on the bench's weights
    d = (got["logits"] - ref["logits"]).abs().max().item()
    record_margin(f"packed_c3_{'vision' if vision else 'text'}/{dt}", packed_vs_padded=d, packed_rows=int(got["packed_rows"]), padded_rows=Bq * K * S)
    if vision:      # measured 1.3e-4: the per-pair vision GEMMs run other tile shapes than the per-query ones, a 16-bit rounding flips
        assert d < 5e-4         # here and there (the fp16 forward itself sits 3.2e-4 from fp32 at this size)
    else:
        assert d == 0.0 and torch.equal(got["order"], ref["order"])


def test_sharded_slices_and_the_whole_list_agree_within_the_parity_gate():
    """a pair's roundings depend on how many rows share its launch — below 128 tiles of 256 x 256 (about 11k rows at
    H = 768) the residual stream between the epilogues is fp32, above it the (hi, lo) pair (rr_gemm_split_ok), and the cut lies
    between the 13-pair and 25-pair strong-scaling shards of one K = 100 query.  So the logits of a sharded run are NOT bit-equal
    to the single-GPU run; what must hold is the parity bar itself: on c3_full (K = 100, S = 512, 81 vision tokens) the eight
    13 / 12-pair slices of SURVEY 8(e) (fp32 stream) and the whole list (split stream) are each within north_star's 1e-3 of the
    fp32 stock-HF golden in fp16, and the pairwise difference — which two results that close to the same reference bound by the
    triangle inequality — is recorded (measured ~2e-4), not gated by a looser number of its own."""
    import rmr_amd
    cfg, w, vision, qs = load_fullsize("c3_full")
    q = qs[0]
    eng = _engine(cfg, vision, w, "fp16")
    K = q["ids"].shape[0]
    ids, am, tt = q["ids"].cuda(), q["am"].cuda(), q["tt"].cuda()
    cls, pat = q["img"][0].cuda(), q["img"][1].cuda()
    whole = eng.forward_ids(ids, am, tt, 1, K, cls, pat, None)["logits"].cpu().reshape(-1)
    parts = torch.full((K,), float("nan"))
    for r in range(8):
        b, e = rmr_amd.shard_range(K, r, 8)
        out = eng.forward_ids(ids, am, tt, 1, K, cls, pat, None, pair_range=(b, e), want_loss=False)["logits"].cpu().reshape(-1)
        parts[b:e] = out[b:e]
    torch.cuda.synchronize()
    ref = q["fp32"]
    d_whole, d_parts = (whole - ref).abs().max().item(), (parts - ref).abs().max().item()
    d_pair = (whole - parts).abs().max().item()
    print(f"[c3_full/fp16 sharded 8 x 13/12 pairs] |whole - fp32| {d_whole:.2e}  |slices - fp32| {d_parts:.2e}  |whole - slices| {d_pair:.2e}")
    record_margin("c3_full/fp16/sharded_8_slices", gate=1e-3, whole_vs_slices=d_pair, **margin_stats(parts, ref))
    assert torch.isfinite(parts).all()
    assert d_whole <= 1e-3 and d_parts <= 1e-3        # each against the fp32 golden, north_star's tolerance
