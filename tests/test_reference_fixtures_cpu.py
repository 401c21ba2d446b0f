"""CPU: the oracle (and the product's host-side helpers) against fixtures that were produced by executing the
REFERENCE'S OWN functions / statements in the build container (tests/golden/make_reference_fixtures.py ->
tests/golden/reference_fn.npz).  This is the part of the oracle that is pinned by the reference itself rather than by the
third-party HF modules: heads and losses, mask construction, the joint-sequence assembly, the [query|image|context]
reorder, both attention-fusion biases, the executor's stable descending sort and the Recall/Precision@K accumulation."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, O


@pytest.fixture(scope="module")
def ref():
    z = np.load(os.path.join(GOLDEN, "reference_fn.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_prepare_logits_labels_and_losses(ref):
    for i in range(int(ref["n_head_cases"])):
        p = f"head{i}."
        loss_fn = str(ref[p + "loss_fn"])
        pw = None if np.isnan(ref[p + "pos_weight"]) else float(ref[p + "pos_weight"])
        Bq, K = int(ref[p + "Bq"]), int(ref[p + "K"])
        labels = [float(x) for x in ref[p + "labels_in"]] if ref[p + "labels_in"].size else None
        logits, lab = O.prepare_logits_labels(loss_fn, torch.from_numpy(ref[p + "l1"]), torch.from_numpy(ref[p + "l2"]),
                                              Bq, K - 1, labels)
        assert torch.equal(logits, torch.from_numpy(ref[p + "logits_out"])), p
        assert lab.dtype == torch.from_numpy(ref[p + "labels_out"]).dtype
        assert torch.equal(lab, torch.from_numpy(ref[p + "labels_out"])), p
        loss = O.loss_value(loss_fn, pw, logits, lab)
        assert abs(loss.item() - float(ref[p + "loss"])) <= 1e-7 * max(1.0, abs(float(ref[p + "loss"]))), p


def test_head_error_behaviour(ref):
    assert str(ref["err.labels_with_negative_sampling"]) == "AssertionError"
    assert str(ref["err.labels_not_a_list"]) == "AssertionError"
    assert str(ref["err.unknown_loss"]) == "ValueError"
    with pytest.raises(AssertionError):
        O.prepare_logits_labels("negative_sampling", torch.zeros(4, 1), torch.zeros(4, 1), 2, 1, [0., 1., 0., 0.])
    with pytest.raises(AssertionError):
        O.prepare_logits_labels("BCE", torch.zeros(4, 1), torch.zeros(4, 1), 2, 1, torch.zeros(4))
    with pytest.raises(ValueError):
        O.loss_value("hinge", None, torch.zeros(4, 1), torch.zeros(4, 1))


def test_inverted_attention_mask(ref):
    assert torch.equal(O.extended_mask(torch.from_numpy(ref["inv.mask2"])), torch.from_numpy(ref["inv.out2"]))
    # the 3-D form ([B, Tq, Tk] -> [B, 1, Tq, Tk]) is the same arithmetic with one broadcast axis fewer
    m3 = torch.from_numpy(ref["inv.mask3"])
    assert torch.equal(((1.0 - m3) * O.FMIN)[:, None], torch.from_numpy(ref["inv.out3"]))


def test_query_mask_and_mask(ref):
    ids = torch.from_numpy(ref["qm.ids"])
    instr = int(ref["qm.instruction_token_id"])
    assert torch.equal(O.instruction_query_mask(ids, instr), torch.from_numpy(ref["qm.masked"]))
    assert torch.equal(O.instruction_query_mask(ids, None), torch.from_numpy(ref["qm.plain"]))
    assert torch.equal(O.token_mask(ids), torch.from_numpy(ref["qm.plain"]))
    # the skiplist is empty on this path (rerank_model.py:385-392 passes []); a non-empty one only removes those ids
    skip = set(int(x) for x in ref["qm.skiplist_ids"])
    want = O.token_mask(ids) * torch.tensor([[0.0 if int(x) in skip else 1.0 for x in row] for row in ids.tolist()])
    assert torch.equal(want, torch.from_numpy(ref["qm.skiplist"]))


def test_joint_sequence_assembly(ref):
    K = int(ref["joint.K"])
    j_ids, j_am = O.joint_sequence(torch.from_numpy(ref["joint.query_input_ids"]),
                                   torch.from_numpy(ref["joint.query_attention_mask"]),
                                   torch.from_numpy(ref["joint.context_input_ids"]),
                                   torch.from_numpy(ref["joint.context_attention_mask"]), K)
    assert torch.equal(j_ids, torch.from_numpy(ref["joint.joint_input_ids"]))
    assert torch.equal(j_am, torch.from_numpy(ref["joint.joint_attention_mask"]))
    # images are repeated query-major, K times each (the library broadcasts instead of materialising this)
    rows = ref["joint.pixel_rows"]
    assert all(rows[i] == rows[(i // K) * K] for i in range(len(rows)))


def test_reorder_and_fusion_adjacency(ref):
    ql, S, P = int(ref["fuse.ql"]), int(ref["fuse.S"]), int(ref["fuse.P"])
    x = torch.from_numpy(ref["fuse.inputs"])
    mask = torch.from_numpy(ref["fuse.query_mask"]).squeeze(-1)
    xin, m = O.reorder_query_image_context(x, mask, ql, S)
    assert torch.equal(xin, torch.from_numpy(ref["fuse.reordered_inputs"]))
    assert torch.equal(m, torch.from_numpy(ref["fuse.reordered_mask"]))
    adj = O.fusion_adjacency(torch.from_numpy(ref["fuse.scores"]), ql, P, S, float(ref["fuse.mult"]))
    assert torch.equal(adj, torch.from_numpy(ref["fuse.adj"]))


def test_interaction_fusion_adjacency(ref):
    K = int(ref["ifuse.K"])
    adj = O.interaction_fusion_adjacency(torch.from_numpy(ref["ifuse.scores"]), float(ref["ifuse.mult"]))
    assert torch.equal(adj, torch.from_numpy(ref["ifuse.adj"]))
    q = torch.from_numpy(ref["ifuse.query_li"])
    assert torch.equal(q.repeat_interleave(K, 0), torch.from_numpy(ref["ifuse.query_li_expanded"]))
    assert torch.equal(torch.from_numpy(ref["ifuse.query_mask"]).repeat_interleave(K, 0),
                       torch.from_numpy(ref["ifuse.query_mask_expanded"]))


def test_stable_sort_and_recall_precision(ref):
    import rmr_amd
    logits, order = ref["met.logits"], ref["met.order"]
    Ks = [int(k) for k in ref["met.Ks"]]
    pos, off = [], 0
    for n in ref["met.pos_len"]:
        pos.append([int(v) for v in ref["met.pos_flat"][off: off + int(n)]])
        off += int(n)
    for rank_fn in (O.rank_descending_stable, rmr_amd.rank_descending_stable):
        got = [rank_fn([float(v) for v in row]) for row in logits]
        assert got == order.tolist()                                     # ties keep retrieval order
    raw = [list(range(logits.shape[1]))] * logits.shape[0]
    for fn in (O.recall_precision_at_k, rmr_amd.recall_precision_at_k):
        r = fn(order.tolist(), pos, Ks)
        assert np.allclose(r["recall"], ref["met.recall"], atol=0, rtol=1e-15)
        assert np.allclose(r["precision"], ref["met.precision"], atol=0, rtol=1e-15)
        r0 = fn(raw, pos, Ks)
        assert np.allclose(r0["recall"], ref["met.raw_recall"], atol=0, rtol=1e-15)
        assert np.allclose(r0["precision"], ref["met.raw_precision"], atol=0, rtol=1e-15)


def test_query_and_forward_glue_executed_from_the_reference_text(ref):
    """RerankModel.query (rerank_model.py:345-478) and FullContextRerankModel.forward (:541-590) were executed statement
    by statement over stock HF modules with seeded weights (make_reference_fixtures.py glue_fixture): mask multiply, first
    `cross_attn_len` text states for the mapping network's cross-attention, [text | prefix | mapped patches] concat order,
    L2-normalise, the 128 -> 768 mapping, the attention-mask concat with ones for the vision tokens, both heads, labels and
    loss.  The oracle must reproduce every stored intermediate (fp32, different operation order inside the HF modules:
    1e-5; masks exactly)."""
    import ast as _ast
    kw = _ast.literal_eval(str(ref["glue.cfg_json"]))
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = "BCE"
    w = O.make_weights(cfg, seed=int(ref["glue.weight_seed"]), vision=True)
    wv = O.make_vit_weights(cfg, seed=int(ref["glue.vit_weight_seed"]))
    Bq, K = int(ref["glue.Bq"]), int(ref["glue.K"])
    ids, am, tt = (torch.from_numpy(ref[f"glue.{k}"]) for k in ("input_ids", "attention_mask", "token_type_ids"))
    px = torch.from_numpy(ref["glue.pixel_values"])
    labels = [float(x) for x in ref["glue.labels"]]
    with torch.no_grad():
        cls, pat = O.clip_vision_forward(cfg, wv, px)           # last_hidden_state[:, 0], hidden_states[-2][:, 1:]
        out = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, cls, pat, labels, want_taps=True)
        li, qmask = O.query_stage(cfg, w, ids, am, tt, cls.repeat_interleave(K, 0), pat.repeat_interleave(K, 0))
    want_li = torch.from_numpy(ref["glue.late_interaction_output"])
    assert li.shape == want_li.shape                          # [N, S + prefix_len + n_patches, li_dim]: the concat order fixes the shape
    assert (li - want_li).abs().max().item() < 1e-5
    assert (out.taps["late_interaction"] - want_li).abs().max().item() < 1e-5
    assert torch.equal(qmask.float().reshape(-1), torch.from_numpy(ref["glue.query_mask"]).reshape(-1))
    # query_mask comes from id != 0, NOT from the tokenizer's attention mask (the fixture pads with id 0 where the mask is 0)
    assert torch.equal(qmask.float().reshape(ids.shape), (ids != 0).float())
    rin = torch.nn.functional.linear(want_li, w["cross_encoder_input_mapping.weight"], w["cross_encoder_input_mapping.bias"])
    assert (rin - torch.from_numpy(ref["glue.reranker_inputs"])).abs().max().item() < 1e-5
    P = cfg.prefix_len + cfg.n_patches
    want_mask = torch.from_numpy(ref["glue.reranker_attention_mask"])
    assert want_mask.shape == (Bq * K, ids.shape[1] + P)
    assert torch.equal(want_mask, torch.cat([(ids != 0).float(), torch.ones(Bq * K, P)], 1))
    assert (out.logits.reshape(-1) - torch.from_numpy(ref["glue.logits"]).reshape(-1)).abs().max().item() < 2e-5
    assert abs(out.loss.item() - float(ref["glue.loss"])) < 2e-6


def test_truncation_round_trip_and_pairing_loop(ref):
    """utils.py:131-153 executed from the reference's text with the installed BertTokenizer: the product's Python mirror and
    the native tokenizer (rr_tok_*) must yield the same truncated (query, context) string pairs."""
    import tempfile
    from transformers import BertTokenizer
    from test_pair_tokenizer_cpu import make_vocab
    from rmr_amd.pair_inputs import NativePairTokenizer, truncate_and_pair
    vocab = make_vocab()
    with tempfile.TemporaryDirectory() as d:
        vp = os.path.join(d, "vocab.txt")
        open(vp, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
        tok = BertTokenizer(vp, do_lower_case=True)
    queries, ctxs = [str(x) for x in ref["pfc.queries"]], [str(x) for x in ref["pfc.contexts"]]
    mq, mc, _, dpq = (int(x) for x in ref["pfc.args"])
    want = list(zip((str(x) for x in ref["pfc.pair_queries"]), (str(x) for x in ref["pfc.pair_contexts"])))
    assert truncate_and_pair(queries, ctxs, tok, mq, mc, dpq) == want
    nat = NativePairTokenizer(vocab)
    got = truncate_and_pair(queries, ctxs, types_tokenizer(nat), mq, mc, dpq)
    assert got == want


def types_tokenizer(nat):
    """HF-style encode/decode facade over the native tokenizer (encode with a token budget, decode of ids)."""
    class T:
        def encode(self, text, add_special_tokens=False, max_length=-1, truncation=True):
            assert not add_special_tokens
            return nat.encode(text, max_length)

        def decode(self, ids):
            return nat.decode(ids)
    return T()
