"""fp16 operands with trained-like OUTLIER activations.  Every parity golden uses seeded random-init weights,
whose rows are well scaled; trained BERT checkpoints carry a few hidden dimensions whose LayerNorm gain / bias is 10-100x
the rest, and with the folded LayerNorm the RAW pre-LayerNorm rows travel as fp16 MFMA operands and as the `hi` half of the
residual stream.  Here the LayerNorm gains of three hidden dimensions are scaled x50 / x50 / x300 in every LayerNorm of the
text encoder (residual rows reach |x| ~ 1e3), the fp32 oracle is the reference, and the production fp16 mode (folded
LayerNorm, split residual stream) must stay as close to it as the unfolded fp32-stream build and closer than bf16 operands;
the range guard (rr_activation_range_flag) must stay quiet there and fire when a gain of 3e4 drives rows to the fp16 limit."""
import pytest
import torch

from helpers import arch_from_cfg, load_golden, record_margin
import oracle.rerank_oracle as O

pytestmark = pytest.mark.gpu
DIMS = ((7, 50.0), (300, 50.0), (555, 300.0))


def _outlier_weights(cfg, scales, compensate=True):
    """LayerNorm gain and bias of the dimensions in `scales` multiplied in every LayerNorm of the text encoder; with
    `compensate` the columns of the matrices that CONSUME those LayerNorm outputs (Q / K / V, FFN-up, the 768 -> 128
    projection) are divided by the same factor, as a trained model's are: the residual stream then carries the outliers
    (which is what the 16-bit rows have to hold) while scores and FFN pre-activations stay in the range of the unscaled model —
    without it the attention saturates and the logits become a discontinuous function of every operand rounding."""
    w = O.make_weights(cfg, seed=0, vision=False, hf_init=True)
    enc = "context_text_encoder.bert_model."
    for k in w:
        if not k.startswith("context_text_encoder"):
            continue
        for d, s in scales:
            if k.endswith("LayerNorm.weight"):
                w[k][d] *= s
            elif k.endswith("LayerNorm.bias"):
                w[k][d] += 0.02 * s
            elif compensate and k.endswith(("self.query.weight", "self.key.weight", "self.value.weight", "intermediate.dense.weight",
                                            "context_text_encoder_linear.weight")):
                w[k][:, d] /= s
    assert enc + "embeddings.LayerNorm.weight" in w
    return w


def _engine(cfg, w, dt):
    import rmr_amd
    eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, False, dt))
    eng.load_state_dict(w)
    return eng


def test_outlier_hidden_dimensions_keep_fp16_mode_accurate():
    from rmr_amd import _lib
    lib = _lib.load()
    cfg = load_golden("c2")["cfg"]
    w = _outlier_weights(cfg, DIMS)
    Bq, K, S = 2, 6, 64
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=5)
    ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, want_taps=True)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K)
    res = {}
    for dt in ("fp16", "bf16"):
        eng = _engine(cfg, w, dt)
        res[dt] = eng.forward_ids(*args)["logits"].cpu()
        assert not eng.activation_range_exceeded()
        if dt == "fp16":
            try:                                                    # the unfolded build with fp32 residual rows, same operands
                assert lib.rr_set_tuning(b"ln_fold", 0) == 0 and lib.rr_set_tuning(b"resid_split", 0) == 0
                res["fp16_unfolded"] = eng.forward_ids(*args)["logits"].cpu()
            finally:
                lib.rr_set_tuning(b"ln_fold", 1); lib.rr_set_tuning(b"resid_split", 1)
    gold = ref.logits.reshape(-1)
    d = {k: (v - gold).abs().max().item() for k, v in res.items()}
    print("outlier dims", DIMS, "logit std", gold.std().item(), "max |dlogit| vs fp32 oracle:", d)
    record_margin("outliers_c2_S64/fp16", max_abs=d["fp16"], unfolded=d["fp16_unfolded"], bf16=d["bf16"], ref_std=gold.std().item())
    for v in res.values():
        assert torch.isfinite(v).all()
    assert d["fp16"] <= 1.3 * d["fp16_unfolded"] + 1e-4           # folding + the split stream cost nothing on outlier rows
    assert d["fp16"] <= d["bf16"]                                  # and fp16 operands stay the more accurate 16-bit mode


def test_range_guard_fires_before_the_fp16_limit():
    cfg = load_golden("c2")["cfg"]
    w = _outlier_weights(cfg, ((7, 3.0e4),), compensate=False)
    Bq, K, S = 1, 4, 64
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=5)
    eng = _engine(cfg, w, "fp16")
    assert not eng.activation_range_exceeded()
    eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), Bq, K)
    assert eng.activation_range_exceeded()                         # read and reset
    assert not eng.activation_range_exceeded()


def test_range_error_is_sticky_for_callers_that_never_poll():
    """the range guard used to be a flag the caller had to poll.  Now every forward ends with an asynchronous
    copy of the flag into pinned host memory and every forward begins by looking at it: after a forward whose rows left the fp16
    range, the NEXT forward on the handle is refused (RR_ERR_RANGE -> OverflowError) until the flag is read and reset; a healthy
    engine is unaffected (a second handle of the same process keeps running)."""
    cfg = load_golden("c2")["cfg"]
    bad = _engine(cfg, _outlier_weights(cfg, ((7, 3.0e4),), compensate=False), "fp16")
    good = _engine(cfg, O.make_weights(cfg, seed=0, vision=False), "fp16")
    Bq, K, S = 1, 4, 64
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=5)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K)
    bad.forward_ids(*args)                          # raises the device flag; the copy lands with the stream
    torch.cuda.synchronize()
    with pytest.raises(OverflowError):
        bad.forward_ids(*args)
    with pytest.raises(OverflowError):              # sticky: still refused
        bad.forward_ids(*args)
    r = good.forward_ids(*args)                     # the other handle is untouched
    torch.cuda.synchronize()
    assert torch.isfinite(r["logits"]).all()
    assert bad.activation_range_exceeded()          # read + reset ...
    bad.forward_ids(*args)                          # ... the next forward runs again (and raises the flag again)
    torch.cuda.synchronize()
    with pytest.raises(OverflowError):
        bad.forward_ids(*args)


def test_a_bf16_handle_is_not_refused_for_values_fp16_could_not_hold():
    """the 9e8 sum-of-squares limit belongs to fp16 operand rows.  A bf16 handle has fp32's exponent range: the same
    3e4 outlier that makes the fp16 handle sticky-fail above is an ordinary value there — its flag stays clear and every later
    forward keeps running and reproduces itself; only a NON-FINITE row raises a bf16 handle's flag."""
    cfg = load_golden("c2")["cfg"]
    eng = _engine(cfg, _outlier_weights(cfg, ((7, 3.0e4),), compensate=False), "bf16")
    Bq, K, S = 1, 4, 64
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=5)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K)
    r1 = eng.forward_ids(*args)
    torch.cuda.synchronize()
    r2 = eng.forward_ids(*args)                      # not refused
    r3 = eng.forward_ids(*args)
    torch.cuda.synchronize()
    assert torch.isfinite(r1["logits"]).all() and torch.equal(r1["logits"], r2["logits"]) and torch.equal(r2["logits"], r3["logits"])
    assert not eng.activation_range_exceeded()
    # a non-finite row is still an error, also in bf16
    w = _outlier_weights(cfg, ((7, 3.0e4),), compensate=False)
    key = next(k for k in w if k.endswith("word_embeddings.weight"))
    w[key] = w[key].clone()
    w[key][int(ids[0, 3])] = float("inf")
    nan = _engine(cfg, w, "bf16")
    nan.forward_ids(*args)
    torch.cuda.synchronize()
    with pytest.raises(OverflowError, match="not finite"):
        nan.forward_ids(*args)
