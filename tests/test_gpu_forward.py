"""GPU parity of the whole hot path through the C ABI against (a) the committed golden logits from the
stock-HF assembly (fp32), (b) the stock-HF bf16-autocast logits of the same cases (the reference's actual arithmetic,
tests/golden/autocast.npz), (c) the CPU oracle run with the device's 16-bit rounding points (localises a failure).

Gates (DESIGN.md "Numerics"):
  * compute_dtype = "fp16" (the headline mode): |logit - fp32 golden| <= 1e-3 on every case — north_star's tolerance,
    against the fp32 forward, which no bf16-operand forward meets (the reference's own bf16-mixed forward is 2e-3..1e-2
    from its fp32 forward on these cases; autocast.npz);
  * compute_dtype = "bf16": |logit - fp32 golden| <= max(1e-3, 1.5 x the reference's own autocast drift on that case)
    per case (helpers.bf16_gate) AND, over all bert-base cases together, no more drift than the reference's autocast
    (geometric mean of the per-case ratios <= 1);
  * either mode vs the oracle with the same rounding points: the same per-case bound (two 16-bit forwards with identical
    rounding points still differ by accumulation order; the taps say where a real bug sits).
"""
import math

import numpy as np
import pytest
import torch

from helpers import O, arch_from_cfg, autocast_drift, bf16_gate, golden_inputs, load_golden, margin_stats, record_margin

pytestmark = pytest.mark.gpu

TOL_FP16 = 1e-3


def _engine(cfg, vision, w, compute_dtype="bf16"):
    import rmr_amd
    eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, vision, compute_dtype))
    eng.load_state_dict(w)
    return eng


def _run(name, want_taps=False, compute_dtype="bf16"):
    g = load_golden(name)
    cfg, vision = g["cfg"], g["vision"]
    w = O.make_weights(cfg, seed=0, vision=vision)
    eng = _engine(cfg, vision, w, compute_dtype)
    ids, am, tt, img = golden_inputs(g)
    lab = torch.tensor(g["labels_list"]).cuda() if g["labels_list"] is not None else None
    eng.set_debug(True)
    r = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"],
                        img[0].cuda() if vision else None, img[1].cuda() if vision else None, lab,
                        want_scores=True, want_order=True)
    torch.cuda.synchronize()
    return g, w, eng, r


@pytest.mark.parametrize("name", ["tiny", "tiny_mm", "tiny_2h", "c1", "c2", "c3s"])
def test_forward_matches_golden_and_oracle(name):
    g, w, eng, r = _run(name)
    cfg, Bq, K = g["cfg"], g["Bq"], g["K"]
    logits = r["logits"].cpu()
    assert torch.isfinite(logits).all()
    gold = torch.from_numpy(g["logits"]).reshape(-1)
    d32 = (logits - gold).abs().max().item()
    record_margin(f"{name}/bf16", gate=bf16_gate(name), **margin_stats(logits, gold))
    ids, am, tt, img = golden_inputs(g)
    torch.set_num_threads(8)
    with torch.no_grad(), O.device_rounding() as mm:
        emu = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1], g["labels_list"], mm=mm,
                                     want_taps=True)
    demu = (logits - emu.logits.reshape(-1)).abs().max().item()
    # intermediate taps localise a failure: text encoder output and the normalised late-interaction rows
    S, H = g["S"], cfg.hidden
    th = eng.debug_read("text_hidden", Bq * K * S * H).view(Bq * K, S, H)
    dth = (th - emu.taps[f"text_layer_{cfg.layers - 1}"]).abs().max().item()
    T = emu.taps["late_interaction"].shape[1]
    li = eng.debug_read("late_interaction", Bq * K * T * cfg.li_dim).view(Bq * K, T, cfg.li_dim)
    dli = (li - emu.taps["late_interaction"]).abs().max().item()
    print(f"[{name}] |dlogit| vs same-rounding oracle {demu:.2e}, vs fp32 golden {d32:.2e}; "
          f"text_hidden {dth:.2e}; late_interaction {dli:.2e}")
    ac = autocast_drift(name)[0]
    print(f"[{name}] gate {bf16_gate(name):.2e}; the reference's own bf16-autocast drift on this case {ac:.2e}")
    assert dth < 3e-2 and dli < 1e-2
    assert demu <= 2 * bf16_gate(name)       # device vs CPU emulation: the difference of two draws of the same rounding noise
    assert d32 <= bf16_gate(name)
    # loss follows the logits
    assert abs(r["loss"].item() - emu.loss.item()) < 2e-3
    assert abs(r["loss"].item() - float(g["loss"])) < 1e-2
    # device order == stable descending sort of the device logits (bit-exact integer work)
    want = [O.rank_descending_stable(row) for row in logits.view(Bq, K).tolist()]
    assert r["order"].cpu().tolist() == want
    # scores
    if cfg.loss_fn == "BCE":
        assert torch.allclose(r["scores"].cpu(), torch.sigmoid(logits), atol=1e-6)
    elif cfg.loss_fn == "negative_sampling":
        assert torch.allclose(r["scores"].cpu().view(Bq, K), torch.softmax(logits.view(Bq, K), -1), atol=1e-6)


@pytest.mark.parametrize("name", ["tiny_mm", "c1", "c2", "c3s"])
def test_fp16_operand_mode_is_within_1e3_of_fp32_goldens(name):
    """compute_dtype="fp16": same kernels, fp16 MFMA operands.  Meets the north_star tolerance (1e-3) against the
    fp32 stock-HF logits, which bf16 operands cannot (see DESIGN.md "Numerics")."""
    g, w, eng, r = _run(name, compute_dtype="fp16")
    cfg, Bq, K = g["cfg"], g["Bq"], g["K"]
    logits = r["logits"].cpu()
    gold = torch.from_numpy(g["logits"]).reshape(-1)
    d32 = (logits - gold).abs().max().item()
    ids, am, tt, img = golden_inputs(g)
    torch.set_num_threads(8)
    with torch.no_grad(), O.device_rounding(torch.float16) as mm:
        emu = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1], g["labels_list"], mm=mm)
    demu = (logits - emu.logits.reshape(-1)).abs().max().item()
    print(f"[{name}/fp16] |dlogit| vs fp32 golden {d32:.2e}, vs same-rounding oracle {demu:.2e}")
    record_margin(f"{name}/fp16", gate=TOL_FP16, vs_same_rounding_oracle=demu, **margin_stats(logits, gold))
    assert d32 <= TOL_FP16
    assert demu <= TOL_FP16
    assert abs(r["loss"].item() - float(g["loss"])) < 1e-3
    want = [O.rank_descending_stable(row) for row in logits.view(Bq, K).tolist()]
    assert r["order"].cpu().tolist() == want


def test_bf16_drift_is_not_worse_than_the_references_own_autocast():
    """Aggregate form of the bf16 gate: over the bert-base goldens the device's drift from the fp32 logits must not
    exceed the drift of the reference's own bf16-mixed (autocast) forward — geometric mean of the per-case ratios <= 1.
    (Recall@5 / top-5 parity: tests/test_gpu_parity_fullsize.py on the c3_sep fixture, unconditional.)"""
    ratios = []
    for name in ("c1", "c2", "c3s"):
        g, w, eng, r = _run(name)
        d32 = (r["logits"].cpu() - torch.from_numpy(g["logits"]).reshape(-1)).abs().max().item()
        ac = autocast_drift(name)[0]
        ratios.append(d32 / ac)
        print(f"[{name}] device bf16 drift {d32:.2e} / reference autocast drift {ac:.2e} = {d32 / ac:.2f}")
        del eng
    gm = math.exp(sum(math.log(x) for x in ratios) / len(ratios))
    print(f"geometric mean of the ratios: {gm:.2f}")
    assert gm <= 1.0


def test_pair_slices_compose_to_full_forward():
    """Multi-GPU contract on one GPU: two rr_forward calls on disjoint pair slices + rr_head == one full call."""
    g, w, eng, r = _run("tiny_mm")
    Bq, K = g["Bq"], g["K"]
    N = Bq * K
    ids, am, tt, img = golden_inputs(g)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K, img[0].cuda(), img[1].cuda(), None)
    cut = 2                                                  # splits query 0's candidates across "ranks"
    a = eng.forward_ids(*args, pair_range=(0, cut), want_loss=False)["logits"][:cut]
    b = eng.forward_ids(*args, pair_range=(cut, N), want_loss=False)["logits"][cut:]
    full = torch.cat([a, b])
    torch.cuda.synchronize()
    assert torch.equal(full, r["logits"])                    # same kernels, same inputs: bit-identical
    h = eng.head(full, None, None, Bq, K, want_scores=True)
    torch.cuda.synchronize()
    assert torch.equal(h["order"], r["order"])
    assert h["loss"].item() == r["loss"].item()


def test_repeatable_and_workspace_reuse():
    g, w, eng, r = _run("tiny")
    ids, am, tt, _ = golden_inputs(g)
    r2 = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], labels=torch.tensor(g["labels_list"]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(r2["logits"], r["logits"])
    assert eng.workspace_bytes(g["Bq"] * g["K"], g["S"]) > 0


def test_error_behaviour_mirrors_reference():
    import rmr_amd
    g = load_golden("tiny")
    cfg = g["cfg"]
    w = O.make_weights(cfg, 0, False)
    arch = arch_from_cfg(cfg, False)
    eng = rmr_amd.RerankEngine(arch)
    ids, am, tt, _ = golden_inputs(g)
    with pytest.raises(ValueError):                           # forward before weights
        eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"])
    missing = dict(w)
    missing.pop("reranker.classifier1.weight")
    with pytest.raises(KeyError):
        rmr_amd.RerankEngine(arch).load_state_dict(missing)
    bad = dict(w)
    bad["reranker.classifier1.weight"] = torch.zeros(2, cfg.ce_hidden)
    with pytest.raises(AssertionError):
        rmr_amd.RerankEngine(arch).load_state_dict(bad)
    eng.load_state_dict(dict(w, **{"some.unrelated.key": torch.zeros(3)}))    # strict=False semantics
    with pytest.raises(AssertionError):                       # N != Bq*K  (rerank_model.py:527)
        eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["Bq"] + 1, g["K"])
    long_ids = torch.ones(2, cfg.max_pos + 8, dtype=torch.int64).cuda()
    with pytest.raises(AssertionError):                       # S > max_position_embeddings
        eng.forward_ids(long_ids, long_ids, long_ids * 0, 1, 2)
    with pytest.raises(NotImplementedError):                  # image features into a text_only build
        eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], torch.zeros(g["Bq"], 128).cuda(),
                        torch.zeros(g["Bq"], 9, 128).cuda())
    ns = rmr_amd.RerankEngine(dict(arch, loss_fn="negative_sampling"))
    ns.load_state_dict(w)
    with pytest.raises(ValueError):                           # labels with negative_sampling (utils.py:233)
        ns.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], labels=torch.zeros(g["Bq"] * g["K"]).cuda())
    with pytest.raises(ValueError):
        rmr_amd.RerankEngine(dict(arch, loss_fn="hinge"))
    with pytest.raises(NotImplementedError):
        rmr_amd.RerankEngine(dict(arch, hidden=96, heads=2))


def test_module_interface_returns_loss_and_logits():
    import rmr_amd
    g = load_golden("tiny_mm")
    cfg = g["cfg"]
    w = O.make_weights(cfg, 0, True)
    conf = dict(cross_encoder_num_hidden_layers=cfg.ce_layers, cross_encoder_max_position_embeddings=cfg.ce_max_pos,
                loss_fn=cfg.loss_fn, pos_weight=None, max_query_length=8, max_decoder_source_length=g["S"],
                arch=arch_from_cfg(cfg, True))
    m = rmr_amd.FullContextRerankModel(conf, state_dict=w)
    ids, am, tt, img = golden_inputs(g)
    out = m.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["K"] - 1, img[0].cuda(), img[1].cuda())
    assert out.logits.shape == (g["Bq"], g["K"]) and out.loss.dim() == 0      # listwise: [Bq, K]
    assert len(out.logits.squeeze().tolist()) == g["Bq"]                        # caller's .squeeze().tolist()
    assert abs(out.loss.detach().cpu().item() - float(g["loss"])) < 1e-2
    assert list(m.context_vision_encoder.named_parameters()) == []


def test_text_call_signature_native_tokenizer_equals_hf_tokenizer(tmp_path):
    """The reference call signature (texts in, .logits out) with the pair inputs assembled by the library's C++ tokenizer
    and by the HF tokenizer object: same ids, hence bit-identical logits."""
    import os
    import rmr_amd
    from transformers import BertTokenizer
    g = load_golden("tiny")
    cfg = g["cfg"]
    words = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    words += ["what", "is", "the", "color", "of", "this", "bus", "red", "a", "big", "city", "street", "in", "london",
              "double", "decker", ".", ",", "?", "##s", "##es", "buses", "are", "usually", "image", "query"]
    f = os.path.join(tmp_path, "vocab.txt")
    open(f, "w").write("\n".join(words) + "\n")
    tok = BertTokenizer(f, do_lower_case=True)
    assert len(words) <= cfg.vocab_size
    w = O.make_weights(cfg, 0, False)
    base = dict(cross_encoder_num_hidden_layers=cfg.ce_layers, cross_encoder_max_position_embeddings=cfg.ce_max_pos,
                loss_fn="BCE", pos_weight=None, max_query_length=6, max_decoder_source_length=32, text_only=True,
                arch=arch_from_cfg(cfg, False), tokenizer=tok)
    q = ["What is the color of this bus?", "a big city"]
    c = ["London buses are usually red.", "Double decker buses, in London.", "", "this street is big", "the query image", "red"]
    outs = []
    for native in (False, True):
        m = rmr_amd.FullContextRerankModel(dict(base, native_tokenizer=native), state_dict=w)
        outs.append(m(query_text_sequences=q, query_pixel_values=None, context_text_sequences=c, num_negative_examples=2,
                      labels=[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]))
    assert outs[0].logits.shape == (6, 1)
    assert torch.equal(outs[0].logits, outs[1].logits) and outs[0].loss.item() == outs[1].loss.item()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_rerank_model_ids_signature(dtype):
    """RerankModel.forward (ids call signature): joint sequence, instruction masking, [query|image|context] reorder,
    2-head logits and the reference's loss_fn(logits, logits) quirk — against the stock-HF golden."""
    import ast, os
    import numpy as np
    import rmr_amd
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "rm_tiny.npz"), allow_pickle=False)
    cfg = O.OracleConfig(**ast.literal_eval(str(z["cfg_json"])))
    cfg.loss_fn = "2H_BCE"
    w = O.make_weights(cfg, seed=0, vision=True)
    conf = dict(cross_encoder_num_hidden_layers=cfg.ce_layers, cross_encoder_max_position_embeddings=cfg.ce_max_pos,
                loss_fn="2H_BCE", pos_weight=cfg.pos_weight, instruction_token_id=int(z["instruction_token_id"]),
                arch=arch_from_cfg(cfg, True, dtype))
    m = rmr_amd.RerankModel(conf, state_dict=w)
    t = lambda k: torch.from_numpy(z[k]).cuda()
    K = int(z["K"])
    out = m(t("query_input_ids"), t("query_attention_mask"), None, t("context_input_ids"), t("context_attention_mask"),
            K - 1, image_features=(t("image_cls"), t("image_patches")), want_order=True)
    torch.cuda.synchronize()
    d = (out.logits.cpu() - torch.from_numpy(z["logits"])).abs().max().item()
    print(f"[rm_tiny/{dtype}] |dlogit| vs fp32 golden {d:.2e}, loss {out.loss.item():.6f} vs {float(z['loss']):.6f}")
    assert d < (1e-3 if dtype == "fp16" else 2e-3)
    assert abs(out.loss.item() - float(z["loss"])) < 2e-3
    assert out.logits.shape == (int(z["Bq"]) * K, 1)
    with pytest.raises(NotImplementedError):
        m(t("query_input_ids"), t("query_attention_mask"), None, t("context_input_ids"), t("context_attention_mask"), K - 1)
    with pytest.raises(AssertionError):                      # preflmr_scores must be [N, S, ql + image tokens]
        m(t("query_input_ids"), t("query_attention_mask"), None, t("context_input_ids"), t("context_attention_mask"), K - 1,
          image_features=(t("image_cls"), t("image_patches")), preflmr_scores=torch.zeros(1))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_rerank_model_attention_fusion(dtype):
    """RerankModel.forward with `preflmr_scores` (PreFLMR attention fusion, rerank_model.py:276-319): the additive
    [T, T] bias built on the device and applied inside the attention kernel, against the stock-HF golden (BertEncoder
    fed the additive 4-D mask) and the same-rounding oracle; the bias moves these logits by 7e-3, 30x the tolerance."""
    import ast, os
    import numpy as np
    import rmr_amd
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "rm_fuse_tiny.npz"), allow_pickle=False)
    z0 = np.load(os.path.join(GOLDEN, "rm_tiny.npz"), allow_pickle=False)
    cfg = O.OracleConfig(**ast.literal_eval(str(z["cfg_json"])))
    cfg.loss_fn = "2H_BCE"
    w = O.make_weights(cfg, seed=0, vision=True)
    conf = dict(cross_encoder_num_hidden_layers=cfg.ce_layers, cross_encoder_max_position_embeddings=cfg.ce_max_pos,
                loss_fn="2H_BCE", pos_weight=cfg.pos_weight, instruction_token_id=int(z["instruction_token_id"]),
                arch=arch_from_cfg(cfg, True, dtype))
    m = rmr_amd.RerankModel(conf, state_dict=w)
    t = lambda k: torch.from_numpy(z[k])
    K, mult = int(z["K"]), float(z["fusion_multiplier"])
    args = (t("query_input_ids").cuda(), t("query_attention_mask").cuda(), None, t("context_input_ids").cuda(),
            t("context_attention_mask").cuda(), K - 1)
    out = m(*args, preflmr_scores=t("preflmr_scores").cuda(), fusion_multiplier=mult,
            image_features=(t("image_cls").cuda(), t("image_patches").cuda()))
    torch.cuda.synchronize()
    gold = torch.from_numpy(z["logits"])
    d = (out.logits.cpu() - gold).abs().max().item()
    with torch.no_grad(), O.device_rounding(torch.bfloat16 if dtype == "bf16" else torch.float16) as mm:
        emu = O.rerank_model_forward(cfg, w, t("query_input_ids"), t("query_attention_mask"), t("context_input_ids"),
                                     t("context_attention_mask"), K, t("image_cls"), t("image_patches"),
                                     int(z["instruction_token_id"]), mm=mm, preflmr_scores=t("preflmr_scores"),
                                     fusion_multiplier=mult)
    demu = (out.logits.cpu() - emu.logits).abs().max().item()
    effect = (gold - torch.from_numpy(z0["logits"])).abs().max().item()
    print(f"[rm_fuse_tiny/{dtype}] |dlogit| vs fp32 golden {d:.2e}, vs same-rounding oracle {demu:.2e}; fusion moves the logits by {effect:.2e}")
    assert effect > 5e-3
    assert d < (2.5e-4 if dtype == "fp16" else 1e-3) and demu < 2.5e-4
    assert abs(out.loss.item() - float(z["loss"])) < 2e-3
    # pair slices compose (the bias rows follow the slice)
    Bq = int(z["Bq"])
    N = Bq * K
    eng = m.engine
    ql, S = z["query_input_ids"].shape[1], z["context_input_ids"].shape[1]
    q_ids = t("query_input_ids").cuda().repeat_interleave(K, 0)
    q_am = t("query_attention_mask").cuda().repeat_interleave(K, 0)
    j_ids = torch.cat([q_ids, t("context_input_ids").cuda()[:, 2:2 - ql]], 1).contiguous()
    j_am = torch.cat([q_am, t("context_attention_mask").cuda()[:, 2:2 - ql]], 1).contiguous()
    kw = dict(preflmr_scores=t("preflmr_scores").cuda(), fusion_multiplier=mult, want_loss=False)
    parts = [eng.forward_joint(j_ids, j_am, Bq, K, ql, t("image_cls").cuda(), t("image_patches").cuda(),
                               int(z["instruction_token_id"]), pair_range=r, **kw)["logits"][r[0]:r[1]] for r in ((0, 2), (2, N))]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts), out.logits.view(-1))


def test_batched_rerank_loop_on_device_orders():
    """§8f-1: several queries per forward, device-side rank, reference-schema records, Recall@K — end to end on the GPU
    against the same loop driven by the CPU oracle."""
    import rmr_amd
    g = load_golden("tiny")
    cfg = g["cfg"]
    cfg.loss_fn = "negative_sampling"
    w = O.make_weights(cfg, 0, False)
    eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, False, "fp16"))
    eng.load_state_dict(w)
    nq, K, S = 5, 6, g["S"]
    ids, am, tt = O.make_pair_batch(cfg, nq, K, S, seed=77)
    queries = [{"question_id": q, "retrieved_docs": [{"passage_id": f"d{q}_{k}", "content": ""} for k in range(K)],
                "pos_item_ids": [f"d{q}_{(2 * q + 1) % K}"], "rows": slice(q * K, (q + 1) * K)} for q in range(nq)]

    def fwd_gpu(batch):
        rows = torch.cat([torch.arange(q["rows"].start, q["rows"].stop) for q in batch])
        r = eng.forward_ids(ids[rows].cuda(), am[rows].cuda(), tt[rows].cuda(), len(batch), K, want_order=True)
        return {"logits": r["logits"].view(len(batch), K).cpu(), "order": r["order"].cpu(), "loss": r["loss"].item()}

    def fwd_oracle(batch):
        rows = torch.cat([torch.arange(q["rows"].start, q["rows"].stop) for q in batch])
        with torch.no_grad():
            o = O.full_context_forward(cfg, w, ids[rows], am[rows], tt[rows], len(batch), K)
        lg = o.logits.view(len(batch), K)
        return {"logits": lg, "order": [O.rank_descending_stable(x) for x in lg.tolist()], "loss": o.loss.item()}

    a = rmr_amd.rerank_dataset(queries, fwd_gpu, batch_queries=2, Ks=[1, 3, 6], docs_to_rerank=6)
    b = rmr_amd.rerank_dataset(queries, fwd_oracle, batch_queries=2, Ks=[1, 3, 6], docs_to_rerank=6)
    for ra, rb in zip(a["output"], b["output"]):
        sa = [p["score"] for p in ra["top_ranking_passages"]]
        sb = [p["score"] for p in rb["top_ranking_passages"]]
        assert max(abs(x - y) for x, y in zip(sorted(sa), sorted(sb))) < 1e-3
        gaps = [sb[i] - sb[i + 1] for i in range(K - 1)]
        if min(gaps) > 2e-3:      # unambiguous oracle ranking -> identical ranking on the device
            assert [p["passage_id"] for p in ra["top_ranking_passages"]] == [p["passage_id"] for p in rb["top_ranking_passages"]]
    assert a["metrics"]["pos_item_ids_recall_at_6"] == 1.0 == b["metrics"]["pos_item_ids_recall_at_6"]
    assert a["metrics"]["pos_item_ids_raw_recall_at_1"] == b["metrics"]["pos_item_ids_raw_recall_at_1"]


def test_bert_large_shape_text_only():
    """BASELINE configs[4] shape (bert-large: 24 layers, hidden 1024, 16 heads, FFN 4096, cross encoder of the same
    width), text-only, at a size the CPU oracle finishes in seconds (K = 4, S = 128); fp16 operands so that the
    north_star tolerance (1e-3 against the fp32 forward) applies, plus the bf16 default against the same-rounding oracle."""
    import rmr_amd
    cfg = O.OracleConfig(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16,
                         ce_intermediate=4096, ce_layers=1, ce_max_pos=512)
    cfg.loss_fn = "BCE"
    w = O.make_weights(cfg, seed=0, vision=False, hf_init=True)
    Bq, K, S = 2, 4, 128
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=21, regime="realistic")
    with torch.no_grad():
        ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K)
    # Gate (a yardstick, not the value observed): max(1e-3, 1.5 x the drift of an EXACT-arithmetic forward with this
    # design's rounding points) — the oracle's device_rounding emulation, i.e. what any correct implementation with 16-bit operands at
    # these places gets (one draw of the same rounding noise; the factor is bf16_gate's).  It is 8.2e-4 (fp16) / 4.1e-3 (bf16) here:
    # after 25 layers fp16 operands sit AT north_star's 1e-3 (|logit| up to 1.23).  Measured on the device:
    # fp16 1.1e-03 [`profiles/r05_parity_margins.json` "bert_large_shape_K4_S128/fp16" "max_abs"], bf16 3.3e-03 [`profiles/r05_parity_margins.json` "bert_large_shape_K4_S128/bf16" "max_abs"].
    for dtype, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        with torch.no_grad(), O.device_rounding(tdt) as mm:
            emu = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, mm=mm).logits.reshape(-1)
        emu_drift = (emu - ref.logits.reshape(-1)).abs().max().item()
        tol = max(1e-3, 1.5 * emu_drift)
        eng = rmr_amd.RerankEngine(arch_from_cfg(cfg, False, dtype))
        eng.load_state_dict(w)
        r = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), Bq, K, want_order=True)
        torch.cuda.synchronize()
        d = (r["logits"].cpu() - ref.logits.reshape(-1)).abs().max().item()
        print(f"[bert-large shape/{dtype}] |dlogit| vs fp32 oracle {d:.2e} (gate {tol:.2e} = max(1e-3, 1.5 x {emu_drift:.2e}); |logit| max {ref.logits.abs().max().item():.2f})")
        record_margin(f"bert_large_shape_K4_S128/{dtype}", gate=tol, same_rounding_emulation_drift=emu_drift,
                      **margin_stats(r["logits"].cpu(), ref.logits.reshape(-1)))
        assert torch.isfinite(r["logits"]).all() and d <= tol
        assert r["order"].cpu().tolist() == [O.rank_descending_stable(x) for x in r["logits"].view(Bq, K).cpu().tolist()]
        del eng


def test_sharded_forward_under_a_one_rank_rccl_group_equals_the_plain_forward():
    """The multi-GPU path (pair slice -> RCCL all_gather_into_tensor -> rr_head) with the real engine and the real "nccl"
    backend at world size 1: bit-identical to the single call, for the pointwise, the listwise and the two-head variant."""
    import os
    import socket

    import torch.distributed as dist
    from rmr_amd.sharding import sharded_forward
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        for name in ("tiny", "tiny_mm", "tiny_2h"):
            g, w, eng, r = _run(name)
            ids, am, tt, img = golden_inputs(g)
            lab = torch.tensor(g["labels_list"]).cuda() if g["labels_list"] is not None else None
            for _ in range(2):                              # the second call reuses the preallocated gather buffers
                out = sharded_forward(eng, ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"],
                                      img[0].cuda() if g["vision"] else None, img[1].cuda() if g["vision"] else None, lab,
                                      want_scores=True)
                torch.cuda.synchronize()
                assert torch.equal(out["logits"], r["logits"])
                assert torch.equal(out["order"], r["order"])
                assert out["loss"].item() == r["loss"].item()
                assert torch.equal(out["scores"], r["scores"])
    finally:
        dist.destroy_process_group()


def test_reserved_forward_is_capturable_into_a_graph():
    """After rr_reserve the forward neither allocates nor synchronises: it can be captured into a HIP graph on the
    caller's stream and replayed, and the replay reproduces the eager logits bit for bit."""
    g, w, eng, r = _run("tiny_mm")
    ids, am, tt, img = golden_inputs(g)
    args = (ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], img[0].cuda(), img[1].cuda(), None)
    eng.set_debug(False)
    eng.reserve(g["Bq"] * g["K"], g["Bq"], g["S"])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.reserve(g["Bq"] * g["K"], g["Bq"], g["S"])        # the redo-flag buffer is per stream
        eng.forward_ids(*args, want_order=True)               # warm-up on the capture stream (function attributes)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            out = eng.forward_ids(*args, want_order=True)
        out["logits"].zero_()
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(out["logits"], r["logits"])
    assert torch.equal(out["order"], r["order"])


def test_graph_replay_survives_a_later_larger_forward():
    """a graph captured after rr_reserve holds the workspace / redo-flag addresses in its kernel nodes.  A later,
    larger forward on the same handle and stream must not free them: outgrown blocks are retired until rr_destroy.  The
    shape takes the fixed-reference attention schedule (640 pairs x 2 heads = 1 280 workgroups >= 1 024), so the redo-flag
    buffer is part of the capture; the graph is replayed after the handle has grown to twice the batch."""
    g = load_golden("tiny_mm")
    cfg = g["cfg"]
    w = O.make_weights(cfg, seed=0, vision=True)
    eng = _engine(cfg, True, w)
    S = g["S"]

    def batch(Bq, K, seed):
        ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=seed)
        cls, pat = O.make_image_feats(cfg, Bq, seed=seed)
        return (ids.cuda(), am.cuda(), tt.cuda(), Bq, K, cls.cuda(), pat.cuda(), None)
    small, big = batch(8, 80, 5), batch(16, 80, 6)
    eager = eng.forward_ids(*small, want_order=True)
    torch.cuda.synchronize()
    ref_logits, ref_order = eager["logits"].clone(), eager["order"].clone()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.reserve(640, 8, S)
        eng.forward_ids(*small, want_order=True)               # warm-up on the capture stream
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            out = eng.forward_ids(*small, want_order=True)
        out["logits"].zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out["logits"], ref_logits) and torch.equal(out["order"], ref_order)
        r_big = eng.forward_ids(*big, want_order=True)          # outgrows the workspace and the flag buffer of this stream
        torch.cuda.synchronize()
        assert torch.isfinite(r_big["logits"]).all()
        out["logits"].zero_()
        graph.replay()                                          # still computes in the blocks it was captured with
        torch.cuda.synchronize()
        assert torch.equal(out["logits"], ref_logits) and torch.equal(out["order"], ref_order)
        again = eng.forward_ids(*small, want_order=True)        # and the handle itself is intact
        torch.cuda.synchronize()
        assert torch.equal(again["logits"], ref_logits)


@pytest.mark.parametrize("name", ["c2", "tiny_mm"])
def test_length_bucketed_forward_equals_the_padded_forward(name):
    """RerankEngine.forward_ids_bucketed (rr_set_padded_seq_len): pairs grouped by the smallest bucket that holds their real
    length and run at that row length must give the padded forward's logits — bit for bit for a text-only model (same rows,
    padded keys contribute exact zeros, same key tiles), up to fp32 summation order in the cross-encoder's attention when
    vision tokens follow the text (their keys land in other tiles) — and the same loss and rank order."""
    g = load_golden(name)
    cfg, vision = g["cfg"], g["vision"]
    w = O.make_weights(cfg, seed=0, vision=vision)
    eng = _engine(cfg, vision, w, "fp16")
    Bq, K, S = 3, 7, g["S"]
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=31, regime="realistic")
    lens = (ids != 0).sum(1)
    assert lens.min().item() <= S // 2 < lens.max().item()                 # the batch really spans several buckets
    img = O.make_image_feats(cfg, Bq, seed=31) if vision else (None, None)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K, None if img[0] is None else img[0].cuda(), None if img[1] is None else img[1].cuda())
    ref = eng.forward_ids(*args, None, want_order=True)
    buckets = (S // 4, S // 2, 3 * S // 4)
    got = eng.forward_ids_bucketed(*args, None, buckets=buckets, want_order=True)
    torch.cuda.synchronize()
    assert got["bucket_rows"] < Bq * K * S
    if vision:
        assert (got["logits"] - ref["logits"]).abs().max().item() < 5e-5
    else:
        assert torch.equal(got["logits"], ref["logits"])
        assert torch.equal(got["order"], ref["order"])
    assert abs(got["loss"].item() - ref["loss"].item()) < 1e-5
    # and the handle is back to plain positions afterwards
    again = eng.forward_ids(*args, None, want_order=True)
    torch.cuda.synchronize()
    assert torch.equal(again["logits"], ref["logits"])


@pytest.mark.parametrize("name", ["tiny", "tiny_mm", "tiny_2h", "c2", "c3s"])
def test_cls_only_cross_encoder_layer_equals_the_all_rows_layer(name):
    """The cross-encoder's last layer behind its K / V projection runs for the CLS row of every pair only (rr_api.hip
    run_cross_encoder: the classifiers read hidden state [:, 0], utils.py:105-108).  Against the same layer computed for
    every row (rr_set_tuning "ce_cls_only" 0; also what the debug taps use): the same logits up to re-decided 16-bit
    roundings, and both within north_star's 1e-3 of the fp32 stock-HF golden (fp16 operands)."""
    from rmr_amd import _lib
    lib = _lib.load()
    g = load_golden(name)
    cfg, vision = g["cfg"], g["vision"]
    w = O.make_weights(cfg, seed=0, vision=vision)
    eng = _engine(cfg, vision, w, "fp16")
    ids, am, tt, img = golden_inputs(g)
    lab = torch.tensor(g["labels_list"]).cuda() if g["labels_list"] is not None else None
    args = (ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], img[0].cuda() if vision else None, img[1].cuda() if vision else None, lab)
    cls_only = eng.forward_ids(*args, want_order=True)
    try:
        assert lib.rr_set_tuning(b"ce_cls_only", 0) == 0
        all_rows = eng.forward_ids(*args, want_order=True)
    finally:
        lib.rr_set_tuning(b"ce_cls_only", 1)
    torch.cuda.synchronize()
    gold = torch.from_numpy(g["logits"]).reshape(-1)
    a, b = cls_only["logits"].cpu(), all_rows["logits"].cpu()
    d_ab, d_a, d_b = (a - b).abs().max().item(), (a - gold).abs().max().item(), (b - gold).abs().max().item()
    print(f"[{name}/fp16] CLS-only vs all rows {d_ab:.2e}; vs fp32 golden: CLS-only {d_a:.2e}, all rows {d_b:.2e}")
    record_margin(f"{name}/fp16/cls_only_layer", cls_only_vs_all_rows=d_ab, cls_only_vs_golden=d_a, all_rows_vs_golden=d_b)
    assert d_ab <= 5e-4 and d_a <= 1e-3 and d_b <= 1e-3
    assert abs(cls_only["loss"].item() - all_rows["loss"].item()) < 1e-4
    if cfg.loss_fn == "2H_BCE":
        assert (cls_only["logits2"] - all_rows["logits2"]).abs().max().item() <= 5e-4


def test_two_handles_in_one_process_may_differ_in_their_numerics_options():
    """SURVEY 8(b) "no global state => several handles per process" (VERDICT r3 item 7): what changes the arithmetic of a forward
    is a handle option (rr_set_option).  Two engines on the same weights, one with the all-rows cross-encoder layer, fp32
    residual rows and LayerNorm kernels, the other with the defaults, run INTERLEAVED: each reproduces exactly what a lone engine
    gives under the process-wide diagnostic switch of the same name, neither leaks into the other, and the effective values
    read back."""
    from rmr_amd import _lib
    lib = _lib.load()
    g = load_golden("c2")
    cfg = g["cfg"]
    w = O.make_weights(cfg, seed=0, vision=False)
    ids, am, tt, _ = golden_inputs(g)
    args = (ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], None, None, None)
    pinned = {"ce_cls_only": 0, "resid_split": 0, "ln_fold": 0}
    a, b = _engine(cfg, False, w, "fp16"), _engine(cfg, False, w, "fp16")
    for k, v in pinned.items():
        a.set_option(k, v)
    assert all(a.get_option(k) == v for k, v in pinned.items()) and all(b.get_option(k) == 1 for k in pinned)
    assert a.get_option("attn_fixed_ref") == 3          # unpinned: the EFFECTIVE (process-wide) attention schedule mode, not -1
    a.set_option("attn_fixed_ref", 0)
    assert a.get_option("attn_fixed_ref") == 0 and b.get_option("attn_fixed_ref") == 3
    a.set_option("attn_fixed_ref", -1)
    with pytest.raises(ValueError):
        a.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        a.set_option("ln_fold", 2)
    ra1, rb1, ra2, rb2 = a.forward_ids(*args), b.forward_ids(*args), a.forward_ids(*args), b.forward_ids(*args)
    torch.cuda.synchronize()
    assert torch.equal(ra1["logits"], ra2["logits"]) and torch.equal(rb1["logits"], rb2["logits"])
    assert not torch.equal(ra1["logits"], rb1["logits"])             # different arithmetic (re-decided roundings) ...
    assert (ra1["logits"] - rb1["logits"]).abs().max().item() < 1e-3  # ... of the same function
    lone = _engine(cfg, False, w, "fp16")
    try:
        for k, v in pinned.items():
            assert lib.rr_set_tuning(k.encode(), v) == 0
        assert b.get_option("ln_fold") == 0                            # an unpinned handle follows the process-wide switch
        rl = lone.forward_ids(*args)
        b.set_option("ln_fold", 1); b.set_option("ce_cls_only", 1); b.set_option("resid_split", 1)   # pinned: immune to it
        rb3 = b.forward_ids(*args)
    finally:
        for k in pinned:
            lib.rr_set_tuning(k.encode(), 1)
    torch.cuda.synchronize()
    assert torch.equal(rl["logits"], ra1["logits"])
    assert torch.equal(rb3["logits"], rb1["logits"])
    a.set_option("ce_cls_only", -1); a.set_option("resid_split", -1); a.set_option("ln_fold", -1)
    assert torch.equal(a.forward_ids(*args)["logits"], rb1["logits"])


@pytest.mark.parametrize("S,n", [(4, 1), (3, 1), (2, 3), (6, 2)])
def test_cls_only_layer_on_very_short_rows(S, n):
    """the CLS-only layer's eleven n-row buffers were carved out of the FFN intermediate (n * T * I * 2 bytes) with no
    capacity check: for rows of <= 5-6 tokens they overran it (for S <= 3 past the end of the workspace).  They are a workspace
    region of their own now (rr_workspace_bytes covers it).  Text-only tiny model at S = 2..6, one to three pairs: finite
    logits, equal to the all-rows layer up to re-decided roundings, and intact neighbours (a second, longer forward on the same
    handle still reproduces its own result)."""
    from rmr_amd import _lib
    lib = _lib.load()
    g = load_golden("tiny")
    cfg = g["cfg"]
    w = O.make_weights(cfg, seed=0, vision=False)
    eng = _engine(cfg, False, w, "fp16")
    gen = torch.Generator().manual_seed(3 + S)                # (make_pair_batch needs S >= 5: [CLS] q [SEP] ctx [SEP])
    ids = torch.randint(1000, cfg.vocab_size, (n, S), generator=gen)
    ids[:, 0], ids[:, -1] = 101, 102
    am, tt = torch.ones_like(ids), torch.zeros_like(ids)
    args = (ids.cuda(), am.cuda(), tt.cuda(), n, 1, None, None, None)
    a = eng.forward_ids(*args)
    try:
        assert lib.rr_set_tuning(b"ce_cls_only", 0) == 0
        b = eng.forward_ids(*args)
    finally:
        lib.rr_set_tuning(b"ce_cls_only", 1)
    torch.cuda.synchronize()
    assert torch.isfinite(a["logits"]).all() and torch.isfinite(b["logits"]).all()
    assert (a["logits"] - b["logits"]).abs().max().item() <= 5e-4
    with torch.no_grad(), O.device_rounding(torch.float16) as mm:
        ref = O.full_context_forward(cfg, w, ids, am, tt, n, 1, None, None, mm=mm)
    assert (a["logits"].cpu().view(-1) - ref.logits.view(-1)).abs().max().item() <= 1e-3


def test_bucket_below_the_vision_window_is_refused_and_clamped():
    """with image features the mapping network's cross-attention reads the first cross_attn_len text rows of a pair.
    A bucketed forward at a row length below that (rr_set_padded_seq_len + a shorter rr_forward) silently read fewer rows than
    the padded call; the library now refuses it (RR_ERR_BAD_SHAPE -> AssertionError like the reference's shape asserts) and
    forward_ids_bucketed clamps its bucket sizes, so a user bucket of 8 gives the padded forward's logits."""
    g = load_golden("tiny_mm")
    cfg = g["cfg"]
    w = O.make_weights(cfg, seed=0, vision=True)
    eng = _engine(cfg, True, w, "fp16")
    ids, am, tt, img = golden_inputs(g)
    S = ids.shape[1]
    assert cfg.cross_attn_len < S
    short = max(2, cfg.cross_attn_len // 2)
    ids2 = ids.clone(); am2 = am.clone()
    ids2[:, short - 1:] = 0; am2[:, short - 1:] = 0                      # every pair fits the short bucket
    args = (ids2.cuda(), am2.cuda(), tt.cuda(), g["Bq"], g["K"], img[0].cuda(), img[1].cuda(), None)
    padded = eng.forward_ids(*args)
    bucketed = eng.forward_ids_bucketed(*args, buckets=(short,))
    torch.cuda.synchronize()
    assert (padded["logits"] - bucketed["logits"]).abs().max().item() <= 5e-5
    from rmr_amd import _lib as L
    L.check(eng.lib.rr_set_padded_seq_len(eng.h, S), eng.h, "rr_set_padded_seq_len")
    try:
        with pytest.raises((AssertionError, ValueError, RuntimeError)):
            eng.forward_ids(ids2[:, :short].contiguous().cuda(), am2[:, :short].contiguous().cuda(), tt[:, :short].contiguous().cuda(),
                            g["Bq"], g["K"], img[0].cuda(), img[1].cuda(), None)
    finally:
        L.check(eng.lib.rr_set_padded_seq_len(eng.h, 0), eng.h, "rr_set_padded_seq_len")


@pytest.mark.parametrize("name", ["c2", "tiny_mm", "tiny_2h"])
def test_packed_forward_equals_the_bucketed_and_the_padded_forward(name):
    """RerankEngine.forward_ids_packed (rr_forward_packed): the pairs laid out group after group at their group's row length,
    every GEMM of a layer ONE launch over all rows, attention / embeddings / CLS heads per group.  Must equal the bucketed
    forward on the same groups bit for bit (the same kernels on the same rows, only the launch boundaries move), hence the
    padded forward bit for bit for a text-only model and up to fp32 summation order with vision tokens."""
    g = load_golden(name)
    cfg, vision = g["cfg"], g["vision"]
    w = O.make_weights(cfg, seed=0, vision=vision)
    eng = _engine(cfg, vision, w, "fp16")
    Bq, K, S = 3, 7, g["S"]
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=31, regime="realistic")
    img = O.make_image_feats(cfg, Bq, seed=31) if vision else (None, None)
    args = (ids.cuda(), am.cuda(), tt.cuda(), Bq, K, None if img[0] is None else img[0].cuda(), None if img[1] is None else img[1].cuda())
    ref = eng.forward_ids(*args, None, want_order=True)
    gran = S // 4
    sizes = sorted({min(S, max(int(eng.arch.get("cross_attn_len", 32)) if vision else 1, s_)) for s_ in range(gran, S + gran, gran)})
    bucketed = eng.forward_ids_bucketed(*args, None, buckets=tuple(sizes[:-1]), want_order=True)
    got = eng.forward_ids_packed(*args, None, granule=gran, want_order=True)
    torch.cuda.synchronize()
    assert got["packed_rows"] == bucketed["bucket_rows"] < Bq * K * S
    if vision:      # the per-pair vision GEMMs run over other row counts (other tile shapes): equal up to fp32 summation order
        assert (got["logits"] - bucketed["logits"]).abs().max().item() < 5e-5
        assert (got["logits"] - ref["logits"]).abs().max().item() < 5e-5
    else:
        assert torch.equal(got["logits"], bucketed["logits"])
        assert torch.equal(got["order"], bucketed["order"])
        assert torch.equal(got["logits"], ref["logits"])
        if cfg.loss_fn == "2H_BCE":                       # both heads travel through the packed order and back
            assert torch.equal(got["logits2"], ref["logits2"])
    assert abs(got["loss"].item() - ref["loss"].item()) < 1e-5
    # host-side lengths (what the tokenizer knows) give the same groups as the device-side derivation
    host = eng.forward_ids_packed(*args, None, granule=gran, want_order=True, lengths=((ids != 0) | (am != 0)).long().mul(torch.arange(1, S + 1)).amax(1).tolist())
    torch.cuda.synchronize()
    assert torch.equal(host["logits"], got["logits"]) and host["packed_rows"] == got["packed_rows"]
    # one group only (granule = S) is the plain forward; and the handle is unchanged afterwards
    one = eng.forward_ids_packed(*args, None, granule=S, want_order=True)
    again = eng.forward_ids(*args, None, want_order=True)
    torch.cuda.synchronize()
    assert torch.equal(again["logits"], ref["logits"])
    assert (one["logits"] - ref["logits"]).abs().max().item() < 5e-5 if vision else torch.equal(one["logits"], ref["logits"])


def test_packed_forward_refuses_bad_segment_tables():
    """rr_forward_packed argument checks: status codes, no launch (include/rerank_mi355.h)."""
    import ctypes as C
    from rmr_amd import _lib
    g = load_golden("c2")
    cfg = g["cfg"]
    eng = _engine(cfg, False, O.make_weights(cfg, seed=0, vision=False), "fp16")
    S = g["S"]
    ids = torch.ones(4 * S, dtype=torch.int64, device="cuda")
    out = torch.empty(4, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call(ns, lens, nseg=None):
        sn, sl = (C.c_int32 * len(ns))(*ns), (C.c_int32 * len(lens))(*lens)
        return eng.lib.rr_forward_packed(eng.h, ids.data_ptr(), ids.data_ptr(), None, None, None, len(ns) if nseg is None else nseg,
                                         sn, sl, S, out.data_ptr(), None, st)
    assert call([4], [S]) == 0
    assert call([4], [S + 1]) == _lib.RR_ERR_BAD_SHAPE                     # longer than the padded length
    assert call([0], [S]) == _lib.RR_ERR_BAD_SHAPE                         # empty segment
    assert call([2, 2], [S // 2, 0]) == _lib.RR_ERR_BAD_SHAPE
    assert call([4], [S], nseg=0) == _lib.RR_ERR_BAD_SHAPE
    assert call([4], [S], nseg=65) == _lib.RR_ERR_BAD_SHAPE
    assert eng.lib.rr_forward_packed(eng.h, ids.data_ptr(), ids.data_ptr(), None, None, None, 1, None, None, S, out.data_ptr(), None,
                                     st) == _lib.RR_ERR_BAD_ARG
    torch.cuda.synchronize()


def test_lightning_checkpoint_keys_load_through_the_prefix():
    """Reranker_base_executor.py:351-381 loads `checkpoint['state_dict']` with strict=False: keys carry the executor's
    `reranker.` prefix, tensors may be bf16/fp16 (mixed-precision checkpoints), and unrelated entries (optimizer/metric
    state, other modules) ride along.  load_state_dict(prefix="reranker.") must take exactly the path's tensors."""
    import rmr_amd
    g = load_golden("tiny_mm")
    cfg = g["cfg"]
    w = O.make_weights(cfg, seed=0, vision=True)
    ck = {}
    for i, (k, t) in enumerate(w.items()):
        # 16-bit storage for a third of the matrices: values that are exactly representable, so the logits must not move
        if t.dim() == 2 and i % 3 == 0:
            t16 = t.to(torch.bfloat16 if i % 2 else torch.float16)
            w[k] = t16.float()
            ck["reranker." + k] = t16
        else:
            ck["reranker." + k] = t
    ck["reranker.context_text_encoder.bert_model.pooler.dense.weight"] = torch.zeros(cfg.hidden, cfg.hidden)   # unused by the path
    ck["reranker.context_vision_encoder.vision_model.vision_model.post_layernorm.weight"] = torch.ones(cfg.vision_hidden)
    ck["retriever.something.weight"] = torch.zeros(3)                        # another module of the LightningModule
    ck["loss_fn.pos_weight"] = torch.tensor([2.0])
    arch = arch_from_cfg(cfg, True)
    a = rmr_amd.RerankEngine(arch)
    unexpected = a.load_state_dict(ck, prefix="reranker.")
    assert "retriever.something.weight" in unexpected and "loss_fn.pos_weight" in unexpected
    assert any("pooler" in k for k in unexpected) and len(unexpected) == 4
    b = rmr_amd.RerankEngine(arch)
    b.load_state_dict(w)
    ids, am, tt, img = golden_inputs(g)
    args = (ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], img[0].cuda(), img[1].cuda(), None)
    la, lb = a.forward_ids(*args)["logits"], b.forward_ids(*args)["logits"]
    torch.cuda.synchronize()
    assert torch.equal(la, lb)
    with pytest.raises(KeyError):                            # without the prefix nothing matches: missing weights
        rmr_amd.RerankEngine(arch).load_state_dict(ck)
    m = rmr_amd.FullContextRerankModel(dict(cross_encoder_num_hidden_layers=cfg.ce_layers,
                                            cross_encoder_max_position_embeddings=cfg.ce_max_pos, loss_fn=cfg.loss_fn,
                                            arch=arch))
    m.load_state_dict(ck, prefix="reranker.")
    assert torch.equal(m.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), g["K"] - 1, img[0].cuda(), img[1].cuda()).logits.reshape(-1), lb)


@pytest.mark.parametrize("loss", ["BCE", "negative_sampling", "2H_BCE"])
def test_degenerate_lists_and_pairs(loss):
    """Edge cases of the input contract (the reference has no tests of its own for this path; these are the degenerate shapes its
    executor can produce): a list of ONE candidate per query (K = 1: the listwise softmax over a single logit, Recall trivially 1), a
    candidate that is EMPTY after truncation ([CLS] q [SEP] [SEP] + padding), a pair whose attention mask is all zero behind [CLS]
    (every key but one masked), and identical candidates (ties keep retrieval order).  fp16 against the fp32 oracle with
    north_star's 1e-3; rank == stable descending sort of the device logits."""
    g = load_golden("tiny_mm")
    cfg, vision = g["cfg"], g["vision"]
    cfg.loss_fn = loss
    w = O.make_weights(cfg, seed=0, vision=vision)
    eng = _engine(cfg, vision, w, "fp16")
    S = g["S"]
    for Bq, K in ((3, 1), (2, 4)):
        ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=17, regime="realistic")
        img = O.make_image_feats(cfg, Bq, seed=17) if vision else (None, None)
        N = Bq * K
        # an empty candidate: keep the query span, then [SEP] [SEP], rest padding
        sep = int(ids[0][(ids[0] != 0).nonzero().max()])
        qend = int((tt[0] == 1).nonzero().min()) if (tt[0] == 1).any() else S // 2
        ids[0, qend:] = 0; am[0, qend:] = 0; tt[0, qend:] = 0
        ids[0, qend] = sep; am[0, qend] = 1; tt[0, qend] = 1
        # a pair with nothing but [CLS] visible
        ids[N - 1, 1:] = 0; am[N - 1, 1:] = 0; tt[N - 1, :] = 0
        if K >= 4:
            ids[K + 2], am[K + 2], tt[K + 2] = ids[K + 1].clone(), am[K + 1].clone(), tt[K + 1].clone()      # a tie inside query 1
        labels = [1.0] + [0.0] * (K - 1)
        labels = labels * Bq
        lab = None if loss == "negative_sampling" else torch.tensor(labels).cuda()
        r = eng.forward_ids(ids.cuda(), am.cuda(), tt.cuda(), Bq, K, None if img[0] is None else img[0].cuda(),
                            None if img[1] is None else img[1].cuda(), lab, want_scores=True, want_order=True)
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1], None if loss == "negative_sampling" else labels)
        got = r["logits"].cpu().reshape(-1)
        want = ref.logits.reshape(-1) if loss != "2H_BCE" else ref.logits.reshape(-1)
        d = (got - want).abs().max().item()
        print(f"[degenerate/{loss} Bq={Bq} K={K}] |dlogit| vs fp32 oracle {d:.2e}; loss {r['loss'].item():.5f} vs {ref.loss.item():.5f}")
        assert torch.isfinite(got).all() and d <= TOL_FP16
        assert abs(r["loss"].item() - ref.loss.item()) < 2e-3
        order = r["order"].cpu().tolist()
        assert order == [O.rank_descending_stable(x) for x in got.view(Bq, K).tolist()]
        if K >= 4:
            assert got[K + 2].item() == got[K + 1].item() and order[1].index(1) < order[1].index(2)


def test_profile_of_a_forward_accounts_for_its_device_time():
    """rr_set_profiling / rr_get_profile (what bench.py's roofline and kernel_ms_per_step are read from): one event per launch,
    chained — a launch that directly follows another profiled launch on the stream starts where that one ended, so the
    classes' times add up to the device time of the forward: never more than the host-timed forward, and (nothing else
    runs on the stream) most of it.  A second forward after a reset reports the same launches; other stream work between two
    API calls is not billed to any class."""
    import time
    g = load_golden("c2")
    cfg = g["cfg"]
    w = O.make_weights(cfg, seed=0, vision=False)
    ids, am, tt, _ = golden_inputs(g)
    args = (ids.cuda(), am.cuda(), tt.cuda(), g["Bq"], g["K"], None, None, None)
    eng = _engine(cfg, False, w, "fp16")
    for _ in range(2):
        eng.forward_ids(*args)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.forward_ids(*args)                       # (creates the event pool: not part of the timed forward)
    torch.cuda.synchronize()
    eng.get_profile(reset=True)
    t0 = time.perf_counter()
    eng.forward_ids(*args)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    p1 = eng.get_profile(reset=True)
    total = sum(v["ms"] for v in p1.values())
    launches = {k: v["launches"] for k, v in p1.items()}
    assert launches["gemm"] > 0 and launches["attention"] > 0 and all(v["ms"] >= 0.0 for v in p1.values())
    assert 0.3 * wall_ms < total <= 1.02 * wall_ms, (total, wall_ms)
    # foreign work on the stream between two calls: its time must not show up in the next forward's classes
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        x = x @ x * 1e-3
    eng.forward_ids(*args)
    torch.cuda.synchronize()
    p2 = eng.get_profile(reset=True)
    assert {k: v["launches"] for k, v in p2.items()} == launches
    assert sum(v["ms"] for v in p2.values()) < 2.0 * total + 0.5
    eng.set_profiling(False)
