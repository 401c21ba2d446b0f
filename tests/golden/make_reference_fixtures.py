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
  utils.py:129-167   prepare_full_context_inputs (with the installed HF BertTokenizer on a seeded vocabulary)
                                                              -> rmr_amd.pair_inputs.prepare_full_context_inputs, rr_tok_prepare_pairs
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
        nsu = dict(torch=torch)
        compile_defs([find_def(tree_u, "prepare_full_context_inputs")], nsu, "reference:utils.py:129-167")
        queries = ["What is the color of this bus?", "Café in Tōkyō street, London: red bus or big city buses"]
        ctxs = ["London buses are usually red.", "a big city street " * 30, "日本語 bus", "", "they have (big) buses",
                "x" * 150 + " is not a word; it is usually a query image"]
        enc = nsu["prepare_full_context_inputs"](queries, ctxs, tok, 8, 64 - 8 - 4, 64, 3)
        rec["pfc.queries"], rec["pfc.contexts"] = np.array(queries), np.array(ctxs)
        rec["pfc.args"] = np.array([8, 64 - 8 - 4, 64, 3])
        rec["pfc.input_ids"] = enc["input_ids"].numpy()
        rec["pfc.attention_mask"] = enc["attention_mask"].numpy()
        rec["pfc.token_type_ids"] = enc["token_type_ids"].numpy()
        report.append("utils.py:129-167 prepare_full_context_inputs executed with the installed BertTokenizer")
    except Exception as e:                                       # noqa: BLE001
        report.append(f"utils.py:129-167 prepare_full_context_inputs NOT executed: {type(e).__name__}: {e}")

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
