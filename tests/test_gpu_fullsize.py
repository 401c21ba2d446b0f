"""Size-independent properties at BASELINE.json's full shape (bert-base, K = 100, S = 512, 81 vision tokens), where the
CPU oracle is too slow to be the checker: the production GEMM kernel against the simple one through the whole forward,
candidate-permutation equivariance, duplicate candidates, and the rank being a descending stable sort."""
import pytest
import torch

from helpers import O, arch_from_cfg, fullsize_bf16_gate, record_margin

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import rmr_amd
    from rmr_amd import _lib
    cfg = O.OracleConfig()                                  # monoPreFLMR-B defaults, Lc = 1
    w = O.make_weights(cfg, seed=0, vision=True)
    eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, True, "bf16"))
    eng.load_state_dict(w)
    Bq, K, S = 2, 100, 512
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=11, regime="realistic")
    img = O.make_image_feats(cfg, Bq, seed=11)
    return dict(lib=_lib.load(), eng=eng, Bq=Bq, K=K, ids=ids.cuda(), am=am.cuda(), tt=tt.cuda(),
                cls=img[0].cuda(), pat=img[1].cuda())


def _fwd(s, ids=None, am=None, tt=None, **kw):
    r = s["eng"].forward_ids(s["ids"] if ids is None else ids, s["am"] if am is None else am,
                             s["tt"] if tt is None else tt, s["Bq"], s["K"], s["cls"], s["pat"], None, **kw)
    torch.cuda.synchronize()
    return r


def test_production_gemm_equals_simple_gemm_through_the_forward(setup):
    s = setup
    prod = _fwd(s)["logits"]                                # 200 pairs x 512: the persistent ring, split residual stream
    try:
        # the split residual stream (hi + lo instead of fp32 rows) lives in the ring kernel only: compare the kernels on the
        # fp32 stream, where they must agree to the bit
        assert s["lib"].rr_set_tuning(b"resid_split", 0) == 0
        a = _fwd(s)["logits"]
        assert s["lib"].rr_set_gemm_variant(0) == 0
        b = _fwd(s)["logits"]
    finally:
        s["lib"].rr_set_gemm_variant(-1)
        s["lib"].rr_set_tuning(b"resid_split", 1)
    assert torch.isfinite(a).all() and torch.isfinite(prod).all()
    # same products, same fp32 accumulation order along K, same epilogue expressions: the two kernels agree to the bit
    # (an intermittent 1e-3..8e-3 gap here was lost residual terms from compiler-packed f32 code, build.py)
    assert torch.equal(a, b), f"max |dlogit| {(a - b).abs().max().item():.2e}"
    # the split stream carries 19 (bf16 operands) / 22 (fp16) bits of each residual row instead of 24: the logits move by
    # re-decided 16-bit roundings only.  WHAT each stream owes the reference is gated where a reference exists — on c3_full, each
    # against the fp32 golden (test_gpu_forward.py test_split_residual_stream_costs_no_accuracy; helpers.bf16_gate) — and this
    # shape has no golden, so the pairwise distance is RECORDED, and bounded only by what that gate implies (VERDICT r4 item 8:
    # a yardstick, not the value last observed): two bf16-operand forwards that are each within fullsize_bf16_gate("c3_full") =
    # max(1e-3, 1.5 x the reference's own bf16-autocast drift on that shape) of the same fp32 result lie at most twice that apart.
    d_pair = (prod - a).abs().max().item()
    record_margin("c3_shape/bf16/split_vs_fp32_stream_pairwise", bound=2 * fullsize_bf16_gate("c3_full"), max_abs=d_pair)
    assert d_pair <= 2 * fullsize_bf16_gate("c3_full"), f"split vs fp32 residual stream: {d_pair:.2e}"
    c = _fwd(s)["logits"]
    assert torch.equal(prod, c)                             # and run to run


def test_candidate_permutation_equivariance_and_rank(setup):
    s = setup
    Bq, K = s["Bq"], s["K"]
    base = _fwd(s, want_order=True, want_scores=True)
    g = torch.Generator().manual_seed(3)
    perm = torch.stack([torch.randperm(K, generator=g) + q * K for q in range(Bq)]).reshape(-1).cuda()
    p = _fwd(s, s["ids"][perm], s["am"][perm], s["tt"][perm])
    # pairs are independent: permuting candidates inside a query permutes the logits (same tiles see other rows, so
    # equality is up to accumulation-order-free arithmetic: the kernels are row-independent -> bit-identical)
    assert torch.equal(p["logits"], base["logits"][perm])
    lg = base["logits"].view(Bq, K).cpu()
    order = base["order"].cpu()
    for q in range(Bq):
        assert order[q].tolist() == O.rank_descending_stable(lg[q].tolist())
    assert torch.allclose(base["scores"].cpu(), torch.sigmoid(base["logits"].cpu()), atol=1e-6)


def test_duplicate_candidates_score_identically(setup):
    s = setup
    K = s["K"]
    ids, am, tt = s["ids"].clone(), s["am"].clone(), s["tt"].clone()
    ids[K + 7], am[K + 7], tt[K + 7] = ids[K + 3], am[K + 3], tt[K + 3]     # two equal candidates of query 1
    r = _fwd(s, ids, am, tt, want_order=True)
    lg = r["logits"]
    assert lg[K + 7].item() == lg[K + 3].item()
    o = r["order"][1].cpu().tolist()
    assert o.index(3) < o.index(7)                          # ties keep retrieval order (Reranker_base_executor.py:934-935)
