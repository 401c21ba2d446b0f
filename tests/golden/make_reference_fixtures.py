"""Fixtures produced by EXECUTING THE REFERENCE'S OWN CODE (build container only; never on the GPU box).

The reference's modules cannot be imported here as modules (missing pytorch_lightning / easydict / peft / faiss /
colbert / tkinter and the pinned transformers 4.38.2 — ordinary ImportErrors), but its pure-torch /
pure-numpy functions on the hot path do not need any of that.  This script reads the reference source files AS TEXT at
run time from /root/reference (nothing of them is copied into this repository), takes the function definitions — or, for
code that sits inline in a long `forward`, the contiguous statements of the cited line range — out of the parsed AST,
compiles exactly those nodes and runs them on seeded inputs.  No stand-in is written for any missing library: the names
the extracted code needs are `torch`, `torch.nn`, `torch.nn.functional`, `numpy`, `tqdm`, `logging` (all installed) and,
where the code says `self.<attr>` / `config.<attr>`, a plain namespace carrying the reference config's own field values.

What is pinned (reference file:lines -> oracle function the CPU tests compare against the stored outputs):
  utils.py:228-254   prepare_logits_labels                 -> O.prepare_logits_labels
  utils.py:208-224   initialise_loss_fn (+ the loss call)   -> O.loss_value
  utils.py:256-282   invert_attention_mask                 -> O.extended_mask
  rerank_model.py:481-513  RerankModel.query_mask / mask   -> O.instruction_query_mask / O.token_mask
  rerank_model.py:190-224  joint sequence assembly         -> rmr_amd.model.RerankModel.forward plumbing / O.rerank_model_forward
  rerank_model.py:257-319  [query|image|context] reorder + attention-fusion bias -> O.fusion_adjacency (+ reorder)
  interaction_rerank_model.py:125-145  interaction fusion bias  -> the bias of O.interaction_forward
  metrics_processors.py:828-884  Recall/Precision@K accumulation -> O.recall_precision_at_k / rmr_amd.ranking
  Reranker_base_executor.py:934-935  sorted(zip(docs, logits), reverse=True) -> O.rank_descending_stable
  utils.py:131-153   prepare_full_context_inputs, the truncate-by-round-trip + query-major pairing loop (with the installed HF
                     BertTokenizer on a seeded vocabulary; the closing tokenizer.batch_encode_plus call, :157-165, does not
                     exist in the installed transformers 5.x and is NOT executed) -> rmr_amd.pair_inputs.truncate_and_pair
  rerank_model.py:345-478  RerankModel.query, every statement of the body, and
  rerank_model.py:541-590  FullContextRerankModel.forward from the image repeat to the 2H_BCE slice, with
  utils.py:88-108    CrossEncoder.forward — `self.*` bound to stock HF modules carrying seeded weights (BertModel for the
                     text encoder and for the cross encoder's AttentionFusionBertModel, which is only ever called with
                     attention_adj=None here; BertEncoder(is_decoder, add_cross_attention) for the mapping network;
                     CLIPVisionModel for the vision tower), nn.Linear, and the reference's own FLMRMultiLayerPerceptron
                     (modeling_flmr.py:531-546, executed from the AST).  The `return EasyDict(...)` statements are not
                     executed; their keyword expressions are evaluated one by one
                                                              -> O.query_stage / O.full_context_forward (taps)
Classes that subclass HuggingFace internals (AttentionFusionBertModel, MORES_BertLayer: copies of / calls into the
4.38.2 `BertModel.forward` / `BertAttention.forward` signatures) are attempted as well and reported; they do not run
against the installed transformers 5.x and are NOT adapted.

Usage:  python tests/golden/make_reference_fixtures.py      -> tests/golden/reference_fn.npz (+ a report on stdout)
"""
from __future__ import annotations

import ast
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def parse(rel):
    src = open(os.path.join(REF, rel)).read()
    return ast.parse(src), src


def find_def(tree, name, cls=None):
    """FunctionDef `name` at module level, or inside ClassDef `cls`."""
    scope = tree.body
    if cls is not None:
        scope = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    return next(n for n in scope if isinstance(n, ast.FunctionDef) and n.name == name)


def compile_defs(nodes, ns, filename):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, filename, "exec"), ns)
    return ns


def statements_in_lines(fn: ast.FunctionDef, lo: int, hi: int):
    """Top-level statements of `fn` that lie entirely inside source lines [lo, hi]."""
    out = [s for s in fn.body if s.lineno >= lo and s.end_lineno <= hi]
    assert out, (fn.name, lo, hi)
    return out


def run_statements(stmts, ns, filename):
    mod = ast.Module(body=list(stmts), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, filename, "exec"), ns)
    return ns


def NS(**kw):
    return types.SimpleNamespace(**kw)


def eval_expr(node, ns, filename):
    e = ast.Expression(body=node)
    ast.fix_missing_locations(e)
    return eval(compile(e, filename, "eval"), ns)


def run_body(fn: ast.FunctionDef, ns, filename, lo=None, hi=None):
    """Execute the top-level statements of `fn` (optionally only those inside source lines [lo, hi]) except `return`
    statements; returns the Return node met last (its value is evaluated by the caller, piece by piece)."""
    ret = None
    stmts = []
    for st in fn.body:
        if lo is not None and (st.lineno < lo or st.end_lineno > hi):
            continue
        if isinstance(st, ast.Return):
            ret = st
            continue
        if isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant) and isinstance(st.value.value, str):
            continue                                           # docstring
        stmts.append(st)
    run_statements(stmts, ns, filename)
    return ret


def glue_fixture(rec, report, ns_utils):
    """RerankModel.query and FullContextRerankModel.forward executed from the reference's text."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from oracle import rerank_oracle as O
    import make_golden as MG
    from transformers import CLIPVisionConfig, CLIPVisionModel
    kw = dict(vocab_size=300, hidden=64, layers=2, heads=2, intermediate=128, max_pos=32, li_dim=16, ce_hidden=64, ce_layers=1,
              ce_heads=2, ce_intermediate=128, ce_max_pos=64, vision_hidden=32, prefix_len=2, n_patches=4, map_layers=1,
              cross_attn_len=8, vit_layers=2, vit_heads=2, vit_intermediate=64, vit_image_size=32, vit_patch_size=16)
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = "BCE"
    w, wv = O.make_weights(cfg, seed=3, vision=True), O.make_vit_weights(cfg, seed=5)
    hf = MG.HFAssembly(cfg, w, True)                         # stock BertModel x2 + BertEncoder with the seeded weights
    hc = CLIPVisionConfig(hidden_size=cfg.vision_hidden, intermediate_size=cfg.vit_intermediate, num_hidden_layers=cfg.vit_layers,
                          num_attention_heads=cfg.vit_heads, image_size=cfg.vit_image_size, patch_size=cfg.vit_patch_size,
                          hidden_act="quick_gelu", layer_norm_eps=1e-5, attn_implementation="eager")
    clip = CLIPVisionModel(hc).eval()
    sd = clip.state_dict()
    pref = "vision_model." if next(iter(sd)).startswith("vision_model.") else ""
    clip.load_state_dict({k: wv.get(O.VIT_PREFIX + "." + k[len(pref):], t) for k, t in sd.items()})

    def lin(name, bias=True):
        W = w[name + ".weight"]
        m = nn.Linear(W.shape[1], W.shape[0], bias=bias)
        with torch.no_grad():
            m.weight.copy_(W)
            if bias:
                m.bias.copy_(w[name + ".bias"])
        return m.eval()
    # the reference's own MLP class, executed from modeling_flmr.py's AST
    tree_f, _ = parse("src/models/flmr/models/flmr/modeling_flmr.py")
    nsf = dict(torch=torch, nn=nn)
    compile_defs([next(n for n in tree_f.body if isinstance(n, ast.ClassDef) and n.name == "FLMRMultiLayerPerceptron")], nsf,
                 "reference:modeling_flmr.py:531-546")
    D, PL, Vh = cfg.li_dim, cfg.prefix_len, cfg.vision_hidden
    mlp = nsf["FLMRMultiLayerPerceptron"]((Vh, D * PL // 2, D * PL)).eval()
    with torch.no_grad():
        for i in (0, 2):
            mlp.model[i].weight.copy_(w[f"context_vision_projection.model.{i}.weight"])
            mlp.model[i].bias.copy_(w[f"context_vision_projection.model.{i}.bias"])

    tree, _ = parse("src/models/rerank/rerank_model.py")
    nsm = dict(torch=torch, F=F, logger=logging.getLogger("reference"))
    compile_defs([find_def(tree, "query_mask", "RerankModel"), find_def(tree, "mask", "RerankModel")], nsm, "reference:rerank_model.py")
    query_def = find_def(tree, "query", "RerankModel")
    fwd_def = find_def(tree, "forward", "FullContextRerankModel")
    tree_u, _ = parse("src/models/rerank/utils.py")
    ce_fwd = find_def(tree_u, "forward", "CrossEncoder")

    me = NS(device=torch.device("cpu"), dtype=torch.float32, instruction_token_id=None, mask_instruction=False,
            context_text_encoder=hf.text, context_text_encoder_linear=lin("context_text_encoder_linear", bias=False),
            context_vision_encoder=lambda pixel_values, output_hidden_states=True: clip(pixel_values=pixel_values, output_hidden_states=True),
            context_vision_projection=mlp, late_interaction_embedding_size=D,
            transformer_mapping_input_linear=lin("transformer_mapping_input_linear"),
            transformer_mapping_cross_attention_length=cfg.cross_attn_len, transformer_mapping_network=hf.mapnet,
            transformer_mapping_output_linear=lin("transformer_mapping_output_linear"),
            cross_encoder_input_mapping=lin("cross_encoder_input_mapping"),
            config=NS(loss_fn="BCE", pos_weight=None))
    me.mask = types.MethodType(nsm["mask"], me)
    me.query_mask = types.MethodType(nsm["query_mask"], me)
    me.loss_fn = ns_utils["initialise_loss_fn"](me.config, torch.device("cpu"))

    def query(**kwargs):                                      # every statement of RerankModel.query but its return
        env = dict(torch=torch, self=me, invert_attention_mask=ns_utils["invert_attention_mask"], **kwargs)
        ret = run_body(query_def, env, "reference:rerank_model.py:345-474")
        assert isinstance(ret.value, ast.Call) and getattr(ret.value.func, "id", "") == "EasyDict"
        return NS(**{k.arg: eval_expr(k.value, env, "reference:rerank_model.py:476-478") for k in ret.value.keywords})
    me.query = query

    def bert_model(**kwargs):                                 # AttentionFusionBertModel with attention_adj=None == stock BertModel
        assert kwargs.pop("attention_adj") is None
        return hf.ce(**kwargs)
    ce_self = NS(bert_model=bert_model, classifier1=lin("reranker.classifier1"), classifier2=lin("reranker.classifier2"))

    def reranker(inputs_embeds, attention_mask=None, attention_adj=None, token_type_ids=None):   # CrossEncoder.forward, utils.py:88-108
        env = dict(self=ce_self, inputs_embeds=inputs_embeds, attention_mask=attention_mask, attention_adj=attention_adj,
                   token_type_ids=token_type_ids)
        ret = run_body(ce_fwd, env, "reference:utils.py:88-106")
        return eval_expr(ret.value, env, "reference:utils.py:108")
    me.reranker = reranker

    Bq, K, S = 2, 3, 24
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=9, regime="realistic")
    px = O.make_pixel_values(cfg, Bq, seed=2022)
    labels = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    env = dict(torch=torch, self=me, prepare_logits_labels=ns_utils["prepare_logits_labels"], text_only=False, batch_size=Bq,
               expanded_batch_size=Bq * K, num_negative_examples=K - 1, query_pixel_values=px, labels=labels,
               inputs=NS(input_ids=ids, attention_mask=am, token_type_ids=tt))
    with torch.no_grad():
        run_body(fwd_def, env, "reference:rerank_model.py:541-590", lo=541, hi=590)
    qo = env["query_outputs"]
    rec["glue.cfg_json"] = np.array(repr(kw))
    rec["glue.weight_seed"], rec["glue.vit_weight_seed"], rec["glue.Bq"], rec["glue.K"], rec["glue.S"] = 3, 5, Bq, K, S
    rec["glue.input_ids"], rec["glue.attention_mask"], rec["glue.token_type_ids"] = ids.numpy(), am.numpy(), tt.numpy()
    rec["glue.pixel_values"], rec["glue.labels"] = px.numpy(), np.array(labels, dtype=np.float32)
    rec["glue.late_interaction_output"] = qo.late_interaction_output.numpy()
    rec["glue.pooler_output"] = qo.pooler_output.numpy()
    rec["glue.query_mask"] = qo.query_mask.numpy()
    rec["glue.reranker_inputs"] = env["reranker_inputs"].numpy()
    rec["glue.reranker_attention_mask"] = env["reranker_attention_mask"].numpy()
    rec["glue.logits"], rec["glue.loss"] = env["logits"].numpy(), np.array(env["loss"].item(), dtype=np.float32)
    report.append("rerank_model.py:345-478 (RerankModel.query), :541-590 (FullContextRerankModel.forward) and utils.py:88-108 "
                  "(CrossEncoder.forward) executed over stock HF modules with seeded weights")



def main():
    torch.manual_seed(0)
    rng = np.random.Generator(np.random.PCG64(2022))
    rec = {}
    report = []

    # ------------------------------------------------------------------ utils.py: heads / losses / mask inversion
    tree, _ = parse("src/models/rerank/utils.py")
    ns = dict(torch=torch, nn=nn, F=F, print=lambda *a, **k: None)
    compile_defs([find_def(tree, "initialise_loss_fn"), find_def(tree, "prepare_logits_labels"),
                  find_def(tree, "invert_attention_mask")], ns, "reference:utils.py")
    cases = [("BCE", None, 3, 4, False), ("BCE", 2.0, 3, 4, True), ("BCE", None, 2, 5, True),
             ("2H_BCE", 3.0, 3, 4, False), ("2H_BCE", None, 2, 3, True), ("negative_sampling", None, 4, 6, False)]
    for i, (loss_fn, pw, Bq, K, with_labels) in enumerate(cases):
        cfg = NS(loss_fn=loss_fn, pos_weight=pw)
        l1 = torch.from_numpy(rng.standard_normal((Bq * K, 1)).astype(np.float32))
        l2 = torch.from_numpy(rng.standard_normal((Bq * K, 1)).astype(np.float32))
        labels = [float(x) for x in (rng.random(Bq * K) < 0.3)] if with_labels else None
        logits, lab = ns["prepare_logits_labels"](cfg, l1, l2, Bq, K - 1, labels=labels)
        loss = ns["initialise_loss_fn"](cfg, torch.device("cpu"))(logits, lab)
        p = f"head{i}."
        rec[p + "loss_fn"] = np.array(loss_fn)
        rec[p + "pos_weight"] = np.array(np.nan if pw is None else pw, dtype=np.float32)
        rec[p + "Bq"], rec[p + "K"] = Bq, K
        rec[p + "l1"], rec[p + "l2"] = l1.numpy(), l2.numpy()
        rec[p + "labels_in"] = np.array(labels if labels is not None else [], dtype=np.float32)
        rec[p + "logits_out"], rec[p + "labels_out"] = logits.numpy(), lab.numpy()
        rec[p + "loss"] = np.array(loss.item(), dtype=np.float32)
    rec["n_head_cases"] = len(cases)
    # the error behaviour the shim mirrors
    for what, call in (("labels_with_negative_sampling", lambda: ns["prepare_logits_labels"](
                            NS(loss_fn="negative_sampling", pos_weight=None), torch.zeros(4, 1), torch.zeros(4, 1), 2, 1, labels=[0., 1., 0., 0.])),
                       ("labels_not_a_list", lambda: ns["prepare_logits_labels"](
                            NS(loss_fn="BCE", pos_weight=None), torch.zeros(4, 1), torch.zeros(4, 1), 2, 1, labels=torch.zeros(4))),
                       ("unknown_loss", lambda: ns["initialise_loss_fn"](NS(loss_fn="hinge", pos_weight=None), "cpu"))):
        try:
            call()
            rec["err." + what] = np.array("none")
        except Exception as e:                                   # noqa: BLE001
            rec["err." + what] = np.array(type(e).__name__)
    m2 = torch.from_numpy((rng.random((3, 7)) < 0.7).astype(np.float32))
    m3 = torch.from_numpy((rng.random((2, 4, 5)) < 0.5).astype(np.float32))
    rec["inv.mask2"], rec["inv.out2"] = m2.numpy(), ns["invert_attention_mask"](m2, torch.float32).numpy()
    rec["inv.mask3"], rec["inv.out3"] = m3.numpy(), ns["invert_attention_mask"](m3, torch.float32).numpy()
    report.append("utils.py: prepare_logits_labels, initialise_loss_fn, invert_attention_mask executed")
    global ns_utils_invert
    ns_utils_invert = ns["invert_attention_mask"]          # the reference's own function, for mores_model.py's import of it
    ns_utils_all = ns                                        # prepare_logits_labels / initialise_loss_fn / invert_attention_mask

    # ------------------------------------------------------------------ rerank_model.py: masks
    tree, _ = parse("src/models/rerank/rerank_model.py")
    ns = dict(torch=torch, F=F, logger=logging.getLogger("reference"))
    logging.getLogger("reference").setLevel(logging.CRITICAL)
    compile_defs([find_def(tree, "query_mask", "RerankModel"), find_def(tree, "mask", "RerankModel")], ns,
                 "reference:rerank_model.py")
    instr = 1999
    ids = torch.from_numpy(rng.integers(0, 2000, size=(6, 24)))
    ids[:, 0] = 101
    ids[0, 5] = instr; ids[1, 1] = instr; ids[2, 0] = instr          # separator in the body / at position 1 / at position 0
    ids[3, 9] = instr; ids[3, 15] = instr                             # two separators: the first counts
    ids[4, 18:] = 0                                                   # padding, no separator -> the error branch (sep = 1)
    me = NS(instruction_token_id=instr, mask_instruction=True)
    me.mask = types.MethodType(ns["mask"], me)
    rec["qm.ids"] = ids.numpy()
    rec["qm.instruction_token_id"] = instr
    rec["qm.masked"] = np.array(ns["query_mask"](me, ids, []), dtype=np.float32)
    rec["qm.plain"] = np.array(ns["query_mask"](me, ids, [], mask_instructions=False), dtype=np.float32)
    rec["qm.skiplist"] = np.array(ns["mask"](me, ids, [int(ids[0, 3]), int(ids[1, 4])]), dtype=np.float32)
    rec["qm.skiplist_ids"] = np.array([int(ids[0, 3]), int(ids[1, 4])])
    report.append("rerank_model.py: RerankModel.query_mask, RerankModel.mask executed")

    # ------------------------------------------------------------------ rerank_model.py: joint assembly (:190-224)
    fwd = find_def(tree, "forward", "RerankModel")
    Bq, K, ql, S, P, D = 2, 3, 8, 32, 5, 16
    q_ids = torch.from_numpy(rng.integers(1, 2000, size=(Bq, ql)))
    q_am = torch.ones(Bq, ql, dtype=torch.int64)
    c_ids = torch.from_numpy(rng.integers(1, 2000, size=(Bq * K, S)))
    c_am = torch.from_numpy((rng.random((Bq * K, S)) < 0.8).astype(np.int64))
    px = torch.from_numpy(rng.standard_normal((Bq, 3, 4, 4)).astype(np.float32))
    env = dict(torch=torch, F=F, self=NS(max_position_embeddings=S, device=torch.device("cpu")),
               query_input_ids=q_ids, query_attention_mask=q_am, query_pixel_values=px, context_input_ids=c_ids,
               context_attention_mask=c_am, num_negative_examples=K - 1, batch_size=Bq, expanded_batch_size=Bq * K)
    run_statements(statements_in_lines(fwd, 190, 224), env, "reference:rerank_model.py:190-224")
    rec["joint.query_input_ids"], rec["joint.query_attention_mask"] = q_ids.numpy(), q_am.numpy()
    rec["joint.context_input_ids"], rec["joint.context_attention_mask"] = c_ids.numpy(), c_am.numpy()
    rec["joint.K"] = K
    rec["joint.joint_input_ids"] = env["joint_query_input_ids"].numpy()
    rec["joint.joint_attention_mask"] = env["joint_query_attention_mask"].numpy()
    rec["joint.pixel_rows"] = env["query_pixel_values"][:, 0, 0, 0].numpy()        # repeat_interleave order of the images
    report.append("rerank_model.py:190-224 (joint sequence assembly) executed")

    # ------------------------------------------------------------------ rerank_model.py: reorder + fusion bias (:238-319)
    N, T = Bq * K, S + P
    x = torch.from_numpy(rng.standard_normal((N, T, D)).astype(np.float32))         # cross_encoder_input_mapping output
    qmask = torch.from_numpy((rng.random((N, S, 1)) < 0.85).astype(np.float32))
    scores = torch.from_numpy((3.0 * rng.standard_normal((N, S, ql + P))).astype(np.float32))
    env2 = dict(torch=torch, F=F, self=NS(device=torch.device("cpu")), reranker_inputs=x,
                query_outputs=NS(query_mask=qmask), expanded_batch_size=N, query_text_size=ql, context_text_size=S,
                left_truncate_context_size=2, right_truncate_context_size=2 - ql, preflmr_scores=scores,
                fusion_multiplier=20.0)
    run_statements(statements_in_lines(fwd, 238, 319), env2, "reference:rerank_model.py:238-319")
    rec["fuse.inputs"], rec["fuse.query_mask"], rec["fuse.scores"] = x.numpy(), qmask.numpy(), scores.numpy()
    rec["fuse.ql"], rec["fuse.S"], rec["fuse.P"], rec["fuse.mult"] = ql, S, P, np.float32(20.0)
    rec["fuse.reordered_inputs"] = env2["reranker_inputs"].numpy()
    rec["fuse.reordered_mask"] = env2["reranker_attention_mask"].numpy()
    rec["fuse.adj"] = env2["reranker_attention_adj"].numpy()
    report.append("rerank_model.py:238-319 (vision mask, [query|image|context] reorder, attention-fusion bias) executed")

    # ------------------------------------------------------------------ interaction_rerank_model.py: fusion bias (:125-145)
    tree_i, _ = parse("src/models/rerank/interaction_rerank_model.py")
    fwd_i = find_def(tree_i, "forward", "InteractionRerankModel")
    Lq, Lc = 7, 11
    qli = torch.from_numpy(rng.standard_normal((Bq, Lq, D)).astype(np.float32))
    cli = torch.from_numpy(rng.standard_normal((N, Lc, D)).astype(np.float32))
    qm = torch.from_numpy((rng.random((Bq, Lq)) < 0.9).astype(np.float32))
    sc = torch.from_numpy((2.0 * rng.standard_normal((N, Lc, Lq))).astype(np.float32))
    env3 = dict(torch=torch, F=F, self=NS(device=torch.device("cpu")), query_late_interaction=qli,
                context_late_interaction=cli, query_mask=qm, num_negative_examples=K - 1, expanded_batch_size=N,
                preflmr_scores=sc, fusion_multiplier=5.0)
    run_statements(statements_in_lines(fwd_i, 125, 145), env3, "reference:interaction_rerank_model.py:125-145")
    rec["ifuse.query_li"], rec["ifuse.query_mask"], rec["ifuse.scores"] = qli.numpy(), qm.numpy(), sc.numpy()
    rec["ifuse.K"], rec["ifuse.mult"] = K, np.float32(5.0)
    rec["ifuse.query_li_expanded"] = env3["query_late_interaction"].numpy()
    rec["ifuse.query_mask_expanded"] = env3["query_mask"].numpy()
    rec["ifuse.adj"] = env3["reranker_attention_adj"].numpy()
    report.append("interaction_rerank_model.py:125-145 (query expansion + fusion bias) executed")

    # ------------------------------------------------------------------ metrics_processors.py:828-884 + the executor's sort
    tree_m, _ = parse("src/metrics/metrics_processors.py")
    met = find_def(tree_m, "compute_rerank_DPR_scores_with_pos_ids", "MetricsProcessor")
    from tqdm import tqdm
    Ks = [1, 2, 5, 10]
    nq, Kc = 12, 10
    logits = rng.standard_normal((nq, Kc)).astype(np.float32)
    logits[3, 2] = logits[3, 7]; logits[5, :] = 0.25; logits[8, 0] = logits[8, 1] = logits[8, 9]      # ties: stable sort keeps retrieval order
    pos = [sorted(set(int(v) for v in rng.integers(0, 14, size=int(rng.integers(0, 4))))) for _ in range(nq)]
    pos[0] = []                                                                       # a query without positives
    tree_e, _ = parse("src/executors/Reranker_base_executor.py")
    ev = find_def(tree_e, "evaluate_outputs", "RerankerBaseExecutor")
    # `doc_logits_pairs = list(zip(retrieved_docs, logits_list))` ; `sorted_docs = sorted(doc_logits_pairs, key=..., reverse=True)` (:934-935)
    sort_stmts = sorted((n for n in ast.walk(ev) if isinstance(n, ast.Assign) and n.lineno in (934, 935)), key=lambda n: n.lineno)
    assert [t.targets[0].id for t in sort_stmts] == ["doc_logits_pairs", "sorted_docs"], [ast.dump(t.targets[0]) for t in sort_stmts]
    batch = []
    orders = np.zeros((nq, Kc), dtype=np.int32)
    for qi in range(nq):
        docs = [{"passage_id": i, "content": f"d{i}"} for i in range(Kc)]
        envs = dict(retrieved_docs=docs, logits_list=[float(v) for v in logits[qi]])
        run_statements(sort_stmts, envs, "reference:Reranker_base_executor.py:934-935")
        ranked = [d for d, _ in envs["sorted_docs"]]
        orders[qi] = [d["passage_id"] for d in ranked]
        batch.append({"top_ranking_passages": ranked, "raw_top_ranking_passages": docs, "pos_item_ids": pos[qi]})
    envm = dict(np=np, tqdm=lambda it: it, batch_result=batch, Ks=Ks, max_K=max(Ks), field="pos_item_ids")
    run_statements(statements_in_lines(met, 828, 884), envm, "reference:metrics_processors.py:828-884")
    del tqdm
    rec["met.logits"], rec["met.Ks"], rec["met.order"] = logits, np.array(Ks), orders
    rec["met.pos_flat"] = np.array([p for ps in pos for p in ps], dtype=np.int32)
    rec["met.pos_len"] = np.array([len(ps) for ps in pos], dtype=np.int32)
    for k in ("precision", "recall", "raw_precision", "raw_recall"):
        rec["met." + k] = envm["result"][k]
    report.append("Reranker_base_executor.py:934-935 (stable descending sort) and metrics_processors.py:828-884 executed")

    # ------------------------------------------------------------------ utils.py:129-167 prepare_full_context_inputs
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from test_pair_tokenizer_cpu import make_vocab                       # the vocabulary of the CPU tokenizer tests
        from transformers import BertTokenizer
        import tempfile
        vocab = make_vocab()
        with tempfile.TemporaryDirectory() as d:
            vp = os.path.join(d, "vocab.txt")
            open(vp, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
            tok = BertTokenizer(vp, do_lower_case=True)
        tree_u, _ = parse("src/models/rerank/utils.py")
        pfc = find_def(tree_u, "prepare_full_context_inputs")
        queries = ["What is the color of this bus?", "Café in Tōkyō street, London: red bus or big city buses"]
        ctxs = ["London buses are usually red.", "a big city street " * 30, "日本語 bus", "", "they have (big) buses",
                "x" * 150 + " is not a word; it is usually a query image"]
        envp = dict(query_text_sequences=queries, context_text_sequences=ctxs, tokenizer=tok, max_query_length=8,
                    max_context_length=64 - 8 - 4, max_decoder_source_length=64, docs_per_query=3)
        # :131-153: the two truncate-by-round-trip comprehensions and the query-major pairing loop, as written
        run_statements(statements_in_lines(pfc, 130, 154), envp, "reference:utils.py:131-153")
        pairs = envp["concatenated_sequences"]
        rec["pfc.queries"], rec["pfc.contexts"] = np.array(queries), np.array(ctxs)
        rec["pfc.args"] = np.array([8, 64 - 8 - 4, 64, 3])
        rec["pfc.pair_queries"], rec["pfc.pair_contexts"] = np.array([a for a, _ in pairs]), np.array([b for _, b in pairs])
        report.append("utils.py:131-153 prepare_full_context_inputs: truncation round trips + pairing loop executed with the installed "
                      "BertTokenizer; :157-165 tokenizer.batch_encode_plus NOT executed (absent from transformers 5.x)")
    except Exception as e:                                       # noqa: BLE001
        report.append(f"utils.py:131-153 prepare_full_context_inputs NOT executed: {type(e).__name__}: {e}")

    # ------------------------------------------------------------------ RerankModel.query / FullContextRerankModel.forward glue
    try:
        glue_fixture(rec, report, ns_utils_all)
    except Exception as e:                                       # noqa: BLE001
        import traceback
        traceback.print_exc()
        report.append(f"rerank_model.py query()/forward() glue NOT executed: {type(e).__name__}: {e}")

    # ------------------------------------------------------------------ HF-internal subclasses: attempted, never adapted
    for rel, what in (("src/models/rerank/attention_fusion.py", "AttentionFusionBertModel"),
                      ("src/models/rerank/mores_model.py", "MORESSym")):
        try:
            t, src = parse(rel)
            g = {"__name__": "reference_exec"}
            keep = [n for n in t.body if not (isinstance(n, ast.ImportFrom) and (n.module or "").startswith("src."))]
            exec(compile(ast.fix_missing_locations(ast.Module(body=keep, type_ignores=[])), "reference:" + rel, "exec"), g)
            from transformers import BertConfig
            c = BertConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128)
            c._attn_implementation = "eager"
            m = g[what](c).eval()
            with torch.no_grad():
                if what == "MORESSym":
                    g["invert_attention_mask"] = ns_utils_invert
                    m(torch.zeros(1, 3, 64), torch.zeros(1, 4, 64), torch.ones(1, 3), torch.ones(1, 4))
                else:
                    m(inputs_embeds=torch.zeros(1, 3, 64), attention_mask=torch.ones(1, 3))
            report.append(f"{rel}: {what} RUNS against the installed transformers")
        except Exception as e:                                   # noqa: BLE001
            report.append(f"{rel}: {what} does not run against the installed transformers "
                          f"({type(e).__name__}: {str(e)[:120]}) - not adapted, not used")

    np.savez_compressed(os.path.join(HERE, "reference_fn.npz"), **rec)
    print("\n".join(report))
    print(f"wrote reference_fn.npz with {len(rec)} arrays")


ns_utils_invert = None

if __name__ == "__main__":
    main()
