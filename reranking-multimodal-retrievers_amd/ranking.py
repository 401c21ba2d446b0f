"""Host-side rank + Recall@K bookkeeping the executor performs around the path.

`rank_descending_stable`: Python `sorted(zip(docs, logits), key=score, reverse=True)`
(/root/reference/src/executors/Reranker_base_executor.py:934-935) — ties keep retrieval order.  The device
computes the same order (`order_out` of rr_forward / rr_head); this host version exists for callers that
already hold host logits.
`recall_precision_at_k`: /root/reference/src/metrics/metrics_processors.py:816-890.
"""
from __future__ import annotations

from typing import Dict, List, Sequence


def rank_descending_stable(scores: Sequence[float]) -> List[int]:
    idx = list(range(len(scores)))
    idx.sort(key=lambda i: scores[i], reverse=True)     # list.sort is stable, also with reverse=True
    return idx


def recall_precision_at_k(ranked_ids: Sequence[Sequence], pos_ids: Sequence[Sequence], Ks: Sequence[int]
                          ) -> Dict[str, List[float]]:
    rec = [0.0] * len(Ks)
    prec = [0.0] * len(Ks)
    top = max(Ks)
    for ids, pos in zip(ranked_ids, pos_ids):
        pos = set(pos)
        hits = [1 if p in pos else 0 for p in list(ids)[:top]]
        for j, k in enumerate(Ks):
            h = sum(hits[:k])
            rec[j] += 1.0 if h > 0 else 0.0
            prec[j] += h / k
    n = max(1, len(ranked_ids))
    return {"recall": [r / n for r in rec], "precision": [p / n for p in prec]}
