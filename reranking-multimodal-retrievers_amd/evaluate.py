"""Executor-side rerank loop around the hot path, batched.

The reference reranks ONE query per forward, pulls the K logits to the host (`.squeeze().tolist()`), sorts them in
Python and builds a prediction record (`/root/reference/src/executors/Reranker_base_executor.py:807-976`), then
computes Recall/Precision@K over those records (`src/metrics/metrics_processors.py:816-890`).  Here several queries go
through one forward, the descending stable rank is computed on the device (`order_out`), and only [Bq,K] logits +
[Bq,K] int32 orders cross PCIe once per batch.  Records follow the reference schema so that its offline tools
(`src/tools/rerank_scores.py:72-98`) read them unchanged:

    {"question_id", "top_ranking_passages": [{"passage_id","content","score"}...],
     "raw_top_ranking_passages": [{"passage_id","content","score": None}...],
     "pos_item_ids", "neg_item_ids", "loss"}

`forward_batch(batch_queries) -> dict(logits [Bq,K] tensor-like, order [Bq,K], loss float)` is whatever model call the
caller wires in (e.g. `FullContextRerankModel.forward_ids(..., want_order=True)`), so this module stays host-only.
"""
from __future__ import annotations

import json
from typing import Callable, Dict, Iterable, List, Optional, Sequence

from .ranking import recall_precision_at_k


def build_records(queries: Sequence[dict], logits, order, loss: float) -> List[dict]:
    """One reference-schema record per query.  `queries[i]` = {"question_id", "retrieved_docs": [{"passage_id",
    "content"}, ...] (retrieval order), "pos_item_ids", optional "neg_item_ids"/"question"/"answers"/"gold_answer"}.
    `logits[i][k]` is the score of retrieved doc k, `order[i]` its descending stable rank (device or host)."""
    out = []
    for qi, q in enumerate(queries):
        docs = q["retrieved_docs"]
        row = [float(x) for x in logits[qi]]
        assert len(docs) == len(row), "Length of retrieved_docs and all_logits must match."      # :929-931
        rank = [int(i) for i in order[qi]]
        rec = {
            "question_id": q["question_id"],
            "top_ranking_passages": [{"passage_id": docs[i]["passage_id"], "content": docs[i]["content"],
                                      "score": row[i]} for i in rank],
            "raw_top_ranking_passages": [{"passage_id": d["passage_id"], "content": d["content"], "score": None}
                                         for d in docs],
            "pos_item_ids": list(q["pos_item_ids"]),
            "neg_item_ids": list(q.get("neg_item_ids", [])),
            "loss": float(loss),
        }
        for k in ("answers", "gold_answer", "question"):
            if q.get(k) is not None:
                rec[k] = q[k]
        out.append(rec)
    return out


def compute_rerank_scores(records: Sequence[dict], Ks: Sequence[int], field: str = "pos_item_ids") -> Dict[str, float]:
    """`compute_rerank_DPR_scores_with_pos_ids` (metrics_processors.py:816-890): reranked and raw (pre-rerank)
    recall / precision at every K, keyed exactly like the reference's log dict."""
    pos = [r[field] for r in records]
    new = recall_precision_at_k([[p["passage_id"] for p in r["top_ranking_passages"]] for r in records], pos, Ks)
    raw = recall_precision_at_k([[p["passage_id"] for p in r["raw_top_ranking_passages"]] for r in records], pos, Ks)
    res = {}
    for j, k in enumerate(Ks):
        res[f"{field}_precision_at_{k}"] = new["precision"][j]
        res[f"{field}_recall_at_{k}"] = new["recall"][j]
        res[f"{field}_raw_precision_at_{k}"] = raw["precision"][j]
        res[f"{field}_raw_recall_at_{k}"] = raw["recall"][j]
    return res


def rerank_dataset(queries: Iterable[dict], forward_batch: Callable[[List[dict]], dict], batch_queries: int,
                   Ks: Sequence[int], docs_to_rerank: Optional[int] = None, out_path: Optional[str] = None) -> dict:
    """The evaluate_outputs loop (Reranker_base_executor.py:785-1030), `batch_queries` queries per forward.
    Returns {"metrics": {...}, "output": [records]}; optionally writes the reference's
    `{..}_predictions_rank_{r}.json` payload (`{"output": [...]}`, :1113-1126)."""
    if docs_to_rerank is not None:
        assert docs_to_rerank == max(Ks), "The number of retrieved documents must be equal to the maximum K."   # :806-808
    records: List[dict] = []
    batch: List[dict] = []

    def flush():
        if not batch:
            return
        r = forward_batch(batch)
        logits = r["logits"].tolist() if hasattr(r["logits"], "tolist") else r["logits"]
        order = r["order"].tolist() if hasattr(r["order"], "tolist") else r["order"]
        records.extend(build_records(batch, logits, order, float(r["loss"])))
        batch.clear()

    for q in queries:
        batch.append(q)
        if len(batch) == batch_queries:
            flush()
    flush()
    metrics = compute_rerank_scores(records, Ks)
    metrics["loss"] = sum(r["loss"] for r in records) / max(1, len(records))                      # :1022-1026
    result = {"metrics": metrics, "output": records}
    if out_path:
        with open(out_path, "w") as f:
            json.dump({"output": records}, f, indent=4)
    return result
