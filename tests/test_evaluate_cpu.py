"""CPU: the batched executor-side loop, record schema and Recall@K bookkeeping (Reranker_base_executor.py:785-1030,
metrics_processors.py:816-890) with a stand-in forward."""
import json
import os

import pytest

from helpers import O


def _queries(n, K):
    qs = []
    for i in range(n):
        docs = [{"passage_id": f"p{i}_{k}", "content": f"text {i} {k}"} for k in range(K)]
        qs.append({"question_id": i, "retrieved_docs": docs, "pos_item_ids": [f"p{i}_{(i * 3) % K}"],
                   "question": f"q{i}?"})
    return qs


def test_batched_loop_schema_and_metrics(tmp_path):
    import rmr_amd
    n, K, Ks = 7, 10, [1, 5, 10]
    calls = []

    def forward_batch(batch):
        calls.append(len(batch))
        logits = [[float((q["question_id"] * 7 + k * 3) % 11) for k in range(K)] for q in batch]   # ties on purpose
        order = [O.rank_descending_stable(r) for r in logits]
        return {"logits": logits, "order": order, "loss": 0.25}

    out = str(tmp_path / "test_predictions_rank_0.json")
    res = rmr_amd.rerank_dataset(_queries(n, K), forward_batch, batch_queries=3, Ks=Ks, docs_to_rerank=10, out_path=out)
    assert calls == [3, 3, 1]
    recs = json.load(open(out))["output"]
    assert len(recs) == n and recs == res["output"]
    r0 = recs[2]
    assert set(r0) >= {"question_id", "top_ranking_passages", "raw_top_ranking_passages", "pos_item_ids",
                       "neg_item_ids", "loss", "question"}
    scores = [p["score"] for p in r0["top_ranking_passages"]]
    assert scores == sorted(scores, reverse=True)
    assert [p["passage_id"] for p in r0["raw_top_ranking_passages"]] == [f"p2_{k}" for k in range(K)]
    assert all(p["score"] is None for p in r0["raw_top_ranking_passages"])
    # ties keep retrieval order (stable descending sort, :934-935)
    for r in recs:
        top = r["top_ranking_passages"]
        for a, b in zip(top, top[1:]):
            if a["score"] == b["score"]:
                assert int(a["passage_id"].split("_")[1]) < int(b["passage_id"].split("_")[1])
    # metrics against an independent count
    want_rec = {k: 0 for k in Ks}
    want_raw = {k: 0 for k in Ks}
    for r in recs:
        for k in Ks:
            want_rec[k] += any(p["passage_id"] in r["pos_item_ids"] for p in r["top_ranking_passages"][:k])
            want_raw[k] += any(p["passage_id"] in r["pos_item_ids"] for p in r["raw_top_ranking_passages"][:k])
    for k in Ks:
        assert res["metrics"][f"pos_item_ids_recall_at_{k}"] == pytest.approx(want_rec[k] / n)
        assert res["metrics"][f"pos_item_ids_raw_recall_at_{k}"] == pytest.approx(want_raw[k] / n)
    assert res["metrics"]["pos_item_ids_recall_at_10"] == 1.0 and res["metrics"]["loss"] == 0.25


def test_docs_to_rerank_must_equal_max_k():
    import rmr_amd
    with pytest.raises(AssertionError):
        rmr_amd.rerank_dataset([], lambda b: {}, 2, [5, 10], docs_to_rerank=100)


def test_merge_margins_never_drops_a_key(tmp_path):
    """tools/merge_margins.py (a one-test re-run had been copied over the full parity record): merging a
    partial run refreshes its keys and keeps the others; --fresh with fewer keys and --check with a missing key are refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "merge_margins.py")
    prof = os.path.join(root, "profiles", "rtest_parity_margins.json")
    full = {"a/fp16": {"max_abs": 1e-4, "run": "r1"}, "b/fp16": {"max_abs": 2e-4, "run": "r1"}, "c/bf16": {"max_abs": 3e-3, "run": "r1"}}
    part = {"b/fp16": {"max_abs": 2.5e-4, "run": "r2"}}
    src_full, src_part, want = tmp_path / "full.json", tmp_path / "part.json", tmp_path / "want.json"
    src_full.write_text(json.dumps(full)); src_part.write_text(json.dumps(part)); want.write_text(json.dumps(full))
    try:
        run = lambda *a: subprocess.run([sys.executable, tool, "rtest", *a], capture_output=True, text=True)
        assert run("--src", str(src_full)).returncode == 0
        assert run("--src", str(src_part)).returncode == 0                       # merge: refresh b, keep a and c
        got = json.load(open(prof))
        assert set(got) == set(full) and got["b/fp16"]["run"] == "r2" and got["a/fp16"]["run"] == "r1"
        r = run("--src", str(src_part), "--fresh")                               # would drop a and c
        assert r.returncode != 0 and "refused" in (r.stdout + r.stderr)
        assert set(json.load(open(prof))) == set(full)                           # untouched
        os.remove(prof)
        r = run("--src", str(src_part), "--check", str(want))                    # a, c missing from the result
        assert r.returncode != 0 and "refused" in (r.stdout + r.stderr)
    finally:
        if os.path.exists(prof):
            os.remove(prof)
