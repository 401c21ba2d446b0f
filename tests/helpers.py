"""Shared test helpers: golden loading, oracle<->engine config mapping."""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import rerank_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    kw = ast.literal_eval(str(g["cfg_json"]))
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = str(g["loss_fn"])
    g["cfg"] = cfg
    g["Bq"], g["K"], g["S"] = int(g["Bq"]), int(g["K"]), int(g["S"])
    g["vision"] = bool(g["vision"])
    g["labels_list"] = [float(x) for x in g["labels"]] if g["labels"].size else None
    return g


def golden_inputs(g):
    ids = torch.from_numpy(g["input_ids"])
    am = torch.from_numpy(g["attention_mask"])
    tt = torch.from_numpy(g["token_type_ids"])
    img = (torch.from_numpy(g["image_cls"]), torch.from_numpy(g["image_patches"])) if g["vision"] else (None, None)
    return ids, am, tt, img


def arch_from_cfg(cfg: O.OracleConfig, vision: bool, compute_dtype: str = "bf16") -> dict:
    return dict(vocab_size=cfg.vocab_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                intermediate=cfg.intermediate, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab, ln_eps=cfg.ln_eps,
                li_dim=cfg.li_dim, ce_hidden=cfg.ce_hidden, ce_layers=cfg.ce_layers, ce_heads=cfg.ce_heads,
                ce_intermediate=cfg.ce_intermediate, ce_max_pos=cfg.ce_max_pos, has_vision=int(vision),
                vision_hidden=cfg.vision_hidden, prefix_len=cfg.prefix_len, n_patches=cfg.n_patches,
                map_layers=cfg.map_layers, cross_attn_len=cfg.cross_attn_len, loss_fn=cfg.loss_fn,
                pos_weight=cfg.pos_weight, compute_dtype=compute_dtype)


def autocast_drift(name):
    """max |bf16-autocast logits - fp32 logits| of the stock-HF assembly for a golden case (tests/golden/autocast.npz):
    how far the reference's OWN arithmetic (Lightning precision='bf16') sits from its fp32 forward."""
    z = np.load(os.path.join(GOLDEN, "autocast.npz"), allow_pickle=False)
    return float(z[f"{name}.drift_vs_fp32"]), torch.from_numpy(z[f"{name}.logits"])


def bf16_gate(name):
    """Gate of the bf16-operand mode against the fp32 goldens: 1e-3 (north_star) or, where the reference's own bf16-mixed
    forward is further than that from fp32, 1.5x the reference's drift on the same case (one draw of a ~2e-3 rounding
    noise against another: the factor keeps the gate from being a coin flip; the aggregate over all cases must not exceed
    the reference's, test_bf16_drift_is_not_worse_than_the_references_own_autocast)."""
    return max(1e-3, 1.5 * autocast_drift(name)[0])


def fullsize_bf16_gate(name):
    """bf16_gate for a FULL-SIZE golden (tests/golden/<name>.npz carries the reference's bf16-autocast logits of its own pool):
    max(1e-3, 1.5 x max |autocast - fp32|), the rule of bf16_gate.  Reads the fixture only (no weights are built)."""
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = max(float(np.abs(z[f"q{qi}.pool_logits_autocast"] - z[f"q{qi}.pool_logits"]).max()) for qi in range(int(z["nq"])))
    return max(1e-3, 1.5 * d)


def load_fullsize(name):
    """Full-size goldens (make_golden.py run_fullsize_case): returns (cfg, weights, per-query list of dicts with the
    regenerated pool inputs and the stored fp32 / autocast pool logits)."""
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    kw = ast.literal_eval(str(z["cfg_json"]))
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = "BCE"
    vision, S, pool = bool(z["vision"]), int(z["S"]), int(z["pool"])
    w = O.make_weights(cfg, seed=0, vision=vision, gain=float(z["gain"]) if "gain" in z.files else 1.0)
    qs = []
    for qi in range(int(z["nq"])):
        seed = int(z[f"q{qi}.seed"])
        ids, am, tt = O.make_pair_batch(cfg, 1, pool, S, seed=seed, regime="realistic")
        ids[:, 1:33] = ids[0, 1:33]
        assert int(ids.sum()) == int(z[f"q{qi}.ids_checksum"]), "synthetic input generator drifted"
        img = O.make_image_feats(cfg, 1, seed=seed) if vision else (None, None)
        q = dict(ids=ids, am=am, tt=tt, img=img, fp32=torch.from_numpy(z[f"q{qi}.pool_logits"]),
                 autocast=torch.from_numpy(z[f"q{qi}.pool_logits_autocast"]))
        for k in ("selected", "positive_list_index", "gap_5_6"):
            if f"q{qi}.{k}" in z.files:
                q[k] = z[f"q{qi}.{k}"]
        qs.append(q)
    return cfg, w, vision, qs


RANKING_FIXTURES_C5 = ("c5_sep", "c5_sep_wide", "c5_sep_g20", "c5_sep_g15")      # bert-large ranking fixtures (make_golden.py SEP)


def top5_set(t):
    return set(torch.as_tensor(t).flatten().argsort(descending=True, stable=True)[:5].tolist())


def ranking_yardstick(q):
    """What the REFERENCE's own arithmetic (bf16 autocast, monoPreFLMR-B_pointwise.jsonnet:186,233) does on a ranking fixture's
    list: (selected pool indices, fp32 logits of the list, autocast logits of the list, designed rank-5/6 gap, margin_stats of
    autocast vs fp32, autocast keeps the fp32 top-5 set, RULE BINDS).  The rule of VERDICT r4 item 1(b): a reduced-precision
    mode must keep the fp32 top-5 on every list where the reference's autocast keeps it with max |d| <= gap / 4."""
    sel = torch.from_numpy(q["selected"].astype(np.int64))
    ref, ac = q["fp32"][sel], q["autocast"][sel]
    gap = float(q["gap_5_6"])
    st = margin_stats(ac, ref)
    kept = top5_set(ref) == top5_set(ac)
    # the drift of the reference's arithmetic over the query's whole candidate POOL (300 pairs): the less noisy yardstick for an
    # absolute-drift bound than the maximum over the 200 selected ones
    pool = float((q["autocast"] - q["fp32"]).abs().max())
    return dict(sel=sel, ref=ref, autocast=ac, gap=gap, stats=st, autocast_keeps_top5=bool(kept), pool_autocast_max_abs=pool,
                binds=bool(kept and st["max_abs"] <= gap / 4))


def margin_stats(got, ref):
    """max |d|, max |d - mean d| (what a ranking sees), Spearman rank correlation and top-5 overlap of two logit vectors."""
    got, ref = got.flatten().double(), ref.flatten().double()
    d = got - ref
    rho = torch.corrcoef(torch.stack([ref.argsort().argsort().double(), got.argsort().argsort().double()]))[0, 1].item() \
        if got.numel() > 2 else 1.0
    k = min(5, got.numel())
    top5 = len(set(ref.argsort(descending=True)[:k].tolist()) & set(got.argsort(descending=True)[:k].tolist()))
    return dict(max_abs=float(d.abs().max()), centred=float((d - d.mean()).abs().max()), rho=float(rho), top5=f"{top5}/{k}",
                n=int(got.numel()), ref_std=float(ref.std()) if got.numel() > 1 else 0.0)


_RUN_TAG = None


def record_margin(key, **metrics):
    """Append measured parity margins to a JSON (`pytest -q` prints nothing): $RR_MARGINS_JSON or gpurun_out/parity_margins.json
    under the repo root.  The builder folds it into the committed profiles/rNN_parity_margins.json with tools/merge_margins.py,
    which never drops a key.  Every entry carries `run` = start time + pid of the pytest process that wrote it.
    Never fails a test."""
    import json
    import time
    global _RUN_TAG
    if _RUN_TAG is None:
        _RUN_TAG = time.strftime("%Y%m%dT%H%M%S") + f"-{os.getpid()}"
    path = os.environ.get("RR_MARGINS_JSON", os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        metrics = dict(metrics, run=_RUN_TAG)
        data[key] = metrics
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except Exception:      # noqa: BLE001 — bookkeeping only
        pass
