"""GPU parity of the scoring head (rr_head: loss, scores, descending stable rank) on its own: every loss kind, K from 1 to
the 4096 maximum, ties, +-inf logits, explicit labels — the rank is index work and must match bit for bit."""
import math

import pytest
import torch

from helpers import O, arch_from_cfg

pytestmark = pytest.mark.gpu

TINY = dict(vocab_size=100, hidden=128, layers=1, heads=2, intermediate=128, max_pos=16, ce_hidden=128, ce_heads=2,
            ce_intermediate=128, ce_layers=1, ce_max_pos=32, li_dim=64)


def _engine(loss_fn, pos_weight=None):
    import rmr_amd
    cfg = O.OracleConfig(**TINY)
    cfg.loss_fn, cfg.pos_weight = loss_fn, pos_weight
    return rmr_amd.RerankEngine(arch_from_cfg(cfg, False))          # rr_head needs no weights


def _logits(Bq, K, seed, ties=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Bq, K, generator=g) * 3
    if ties and K >= 8:
        x[:, 3] = x[:, 1]                                   # equal logits: retrieval order decides
        x[:, K - 1] = x[:, 0]
        x[0, 2] = float("-inf")
        x[-1, 5] = float("inf") if K > 5 else x[-1, 0]
    return x


@pytest.mark.parametrize("K", [1, 2, 7, 100, 257, 1000, 4096])
@pytest.mark.parametrize("Bq", [1, 5])
def test_rank_is_descending_and_stable_bit_exact(K, Bq):
    eng = _engine("BCE")
    x = _logits(Bq, K, seed=K + Bq)
    r = eng.head(x.reshape(-1).cuda(), None, None, Bq, K)
    torch.cuda.synchronize()
    want = [O.rank_descending_stable(row) for row in x.tolist()]
    assert r["order"].cpu().tolist() == want
    for row, o in zip(x.tolist(), want):                    # a permutation, sorted, ties in retrieval order
        assert sorted(o) == list(range(K))
        assert all(row[a] > row[b] or (row[a] == row[b] and a < b) for a, b in zip(o, o[1:]))


@pytest.mark.parametrize("pos_weight", [None, 2.5])
@pytest.mark.parametrize("with_labels", [False, True])
def test_bce_head(pos_weight, with_labels):
    eng = _engine("BCE", pos_weight)
    Bq, K = 4, 37
    x = _logits(Bq, K, seed=1, ties=False)
    g = torch.Generator().manual_seed(2)
    labels = (torch.rand(Bq * K, generator=g) < 0.2).float().tolist() if with_labels else None
    r = eng.head(x.reshape(-1).cuda(), None, torch.tensor(labels).cuda() if with_labels else None, Bq, K, want_scores=True)
    torch.cuda.synchronize()
    lg, lab = O.prepare_logits_labels("BCE", x.reshape(-1, 1), x.reshape(-1, 1), Bq, K - 1, labels)
    want = O.loss_value("BCE", pos_weight, lg, lab)
    assert abs(r["loss"].item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    assert torch.allclose(r["scores"].cpu(), torch.sigmoid(x.reshape(-1)), atol=1e-6)


@pytest.mark.parametrize("pos_weight", [None, 3.0])
def test_two_head_ce(pos_weight):
    eng = _engine("2H_BCE", pos_weight)
    Bq, K = 3, 50
    l2, l1 = _logits(Bq, K, seed=3, ties=False), _logits(Bq, K, seed=4, ties=False)     # l2 = ranked (second) head
    r = eng.head(l2.reshape(-1).cuda(), l1.reshape(-1).cuda(), None, Bq, K, want_scores=True)
    torch.cuda.synchronize()
    lg, lab = O.prepare_logits_labels("2H_BCE", l1.reshape(-1, 1), l2.reshape(-1, 1), Bq, K - 1, None)
    want = O.loss_value("2H_BCE", pos_weight, lg, lab)
    assert abs(r["loss"].item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    assert torch.allclose(r["scores"].cpu(), torch.softmax(lg, 1)[:, 1], atol=1e-6)
    assert r["order"].cpu().tolist() == [O.rank_descending_stable(row) for row in l2.tolist()]
    with pytest.raises(ValueError):
        eng.head(l2.reshape(-1).cuda(), None, None, Bq, K)                                # first head missing


@pytest.mark.parametrize("Bq", [6, 64])                       # 64 queries x 100 candidates = BASELINE configs[3]
def test_listwise_head_and_limits(Bq):
    eng = _engine("negative_sampling")
    K = 100
    x = _logits(Bq, K, seed=5, ties=False)
    r = eng.head(x.reshape(-1).cuda(), None, None, Bq, K, want_scores=True)
    torch.cuda.synchronize()
    lg, lab = O.prepare_logits_labels("negative_sampling", x.reshape(-1, 1), x.reshape(-1, 1), Bq, K - 1, None)
    want = O.loss_value("negative_sampling", None, lg, lab)
    assert abs(r["loss"].item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    assert torch.allclose(r["scores"].cpu().view(Bq, K), torch.softmax(x, 1), atol=1e-6)
    with pytest.raises(ValueError):                          # utils.py:233: no labels with the listwise loss
        eng.head(x.reshape(-1).cuda(), None, torch.zeros(Bq * K).cuda(), Bq, K)
    with pytest.raises(NotImplementedError):
        eng.head(torch.zeros(4097).cuda(), None, None, 1, 4097)
    with pytest.raises(AssertionError):
        eng.head(torch.zeros(4).cuda(), None, None, 0, 4)
    # the loss is a fixed-order reduction: identical bits run to run
    a = eng.head(x.reshape(-1).cuda(), None, None, Bq, K)["loss"].item()
    b = eng.head(x.reshape(-1).cuda(), None, None, Bq, K)["loss"].item()
    assert a == b and math.isfinite(a)


def test_device_head_and_rank_against_reference_executed_fixtures():
    """The fixtures of tests/golden/reference_fn.npz were produced by executing the reference's own
    prepare_logits_labels / initialise_loss_fn and its `sorted(zip(docs, logits), reverse=True)`; the device head
    (rr_head) must reproduce the loss values and the exact orders."""
    import os

    import numpy as np

    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "reference_fn.npz"), allow_pickle=False)
    for i in range(int(z["n_head_cases"])):
        p = f"head{i}."
        loss_fn = str(z[p + "loss_fn"])
        pw = None if np.isnan(z[p + "pos_weight"]) else float(z[p + "pos_weight"])
        Bq, K = int(z[p + "Bq"]), int(z[p + "K"])
        eng = _engine(loss_fn, pw)
        l1, l2 = torch.from_numpy(z[p + "l1"]).reshape(-1).cuda(), torch.from_numpy(z[p + "l2"]).reshape(-1).cuda()
        lab = torch.from_numpy(z[p + "labels_in"]).cuda() if z[p + "labels_in"].size else None
        # rr_head takes (ranked logits, first head): classifier1 is head 0 of the 2H_BCE pair, classifier2 the ranked one
        r = eng.head(l2 if loss_fn == "2H_BCE" else l1, l1 if loss_fn == "2H_BCE" else None, lab, Bq, K)
        torch.cuda.synchronize()
        want = float(z[p + "loss"])
        assert abs(r["loss"].item() - want) <= 3e-6 * max(1.0, abs(want)), (p, r["loss"].item(), want)
    logits, order = z["met.logits"], z["met.order"]
    eng = _engine("BCE")
    r = eng.head(torch.from_numpy(logits).reshape(-1).cuda(), None, None, logits.shape[0], logits.shape[1])
    torch.cuda.synchronize()
    assert r["order"].cpu().tolist() == order.tolist()
