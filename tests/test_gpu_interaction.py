"""GPU parity of the Interaction rerankers (ModPreFLMR-BERT = CrossEncoder over cat(query, context) tokens;
ModPreFLMR-IB = MORES) through rr_forward_interaction, against the stock-HF goldens and the same-rounding oracle."""
import ast
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, O, arch_from_cfg, autocast_drift, bf16_gate

pytestmark = pytest.mark.gpu


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    cfg = O.OracleConfig(**ast.literal_eval(str(g["cfg_json"])))
    cfg.loss_fn = str(g["loss_fn"])
    g["cfg"], g["mores"] = cfg, bool(g["mores"])
    for k in ("Bq", "K", "Lq", "Lc"):
        g[k] = int(g[k])
    g["labels_list"] = [float(x) for x in g["labels"]] if g["labels"].size else None
    return g


def _engine(g, compute_dtype="bf16"):
    import rmr_amd
    arch = arch_from_cfg(g["cfg"], False, compute_dtype)
    arch["model_kind"] = "mores" if g["mores"] else "interaction"
    eng = rmr_amd.RerankEngine(arch)
    w = O.make_interaction_weights(g["cfg"], g["mores"], seed=0)
    eng.load_state_dict(w)
    return eng, w


@pytest.mark.parametrize("name", ["int_tiny", "mores_tiny", "int_base", "mores_base"])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_interaction_matches_golden_and_oracle(name, dtype):
    g = _load(name)
    cfg, Bq, K = g["cfg"], g["Bq"], g["K"]
    eng, w = _engine(g, dtype)
    q, c = torch.from_numpy(g["query_li"]), torch.from_numpy(g["context_li"])
    qm, cm = torch.from_numpy(g["query_mask"]), torch.from_numpy(g["context_mask"])
    lab = torch.tensor(g["labels_list"]).cuda() if g["labels_list"] is not None else None
    r = eng.forward_interaction(q.cuda(), c.cuda(), qm.cuda(), cm.cuda(), Bq, K, lab, want_scores=True, want_order=True)
    torch.cuda.synchronize()
    logits = r["logits"].cpu()
    gold = torch.from_numpy(g["logits"]).reshape(-1)
    d32 = (logits - gold).abs().max().item()
    with torch.no_grad(), O.device_rounding(torch.bfloat16 if dtype == "bf16" else torch.float16) as mm:
        emu = O.interaction_forward(cfg, w, q, c, qm, cm, K, g["labels_list"], g["mores"], mm=mm)
    demu = (logits - emu.logits.reshape(-1)).abs().max().item()
    print(f"[{name}/{dtype}] |dlogit| vs fp32 golden {d32:.2e}, vs same-rounding oracle {demu:.2e}")
    assert torch.isfinite(logits).all()
    if dtype == "fp16":
        assert d32 <= 1e-3 and demu <= 1e-3
    else:     # bf16 operands: against the reference's own bf16-autocast drift on the same case (tests/golden/autocast.npz)
        gate = bf16_gate(name)
        print(f"   bf16 gate {gate:.2e}")
        assert d32 <= gate and demu <= 2 * gate     # demu: two draws of the same rounding noise
    assert abs(r["loss"].item() - float(g["loss"])) < 1e-2
    assert r["order"].cpu().tolist() == [O.rank_descending_stable(x) for x in logits.view(Bq, K).tolist()]


def test_module_interface_and_errors():
    import rmr_amd
    g = _load("int_tiny")
    cfg = g["cfg"]
    w = O.make_interaction_weights(cfg, False, seed=0)
    conf = dict(cross_encoder_num_hidden_layers=cfg.ce_layers, cross_encoder_max_position_embeddings=cfg.ce_max_pos,
                loss_fn="BCE", pos_weight=None, interaction_type="NORMAL", arch=arch_from_cfg(cfg, False))
    m = rmr_amd.InteractionRerankModel(conf, state_dict=w)
    q, c = torch.from_numpy(g["query_li"]).cuda(), torch.from_numpy(g["context_li"]).cuda()
    qm, cm = torch.from_numpy(g["query_mask"]).int().cuda(), torch.from_numpy(g["context_mask"]).int().cuda()
    out = m(query_late_interaction=q, context_late_interaction=c, num_negative_examples=g["K"] - 1, query_mask=qm,
            context_mask=cm, labels=g["labels_list"])
    assert out.logits.shape == (g["Bq"] * g["K"], 1) and out.loss.dim() == 0
    assert abs(out.loss.item() - float(g["loss"])) < 1e-2
    with pytest.raises(AssertionError):                   # attention fusion scores must be [N, Lc, Lq]
        m(q, c, g["K"] - 1, qm, cm, preflmr_scores=torch.zeros(1))
    with pytest.raises(AssertionError):
        m(q, c, g["K"], qm, cm)
    with pytest.raises(ValueError):                       # full-context entry point on an interaction handle
        m.engine.forward_ids(torch.ones(2, 8, dtype=torch.int64).cuda(), torch.ones(2, 8, dtype=torch.int64).cuda(),
                             None, 1, 2)


def test_interaction_pair_slices_compose():
    g = _load("mores_tiny")
    eng, w = _engine(g)
    Bq, K = g["Bq"], g["K"]
    N = Bq * K
    args = [torch.from_numpy(g[k]).cuda() for k in ("query_li", "context_li", "query_mask", "context_mask")]
    full = eng.forward_interaction(*args, Bq, K, None, want_order=True)
    a = eng.forward_interaction(*args, Bq, K, None, pair_range=(0, 2), want_loss=False)["logits"][:2]
    b = eng.forward_interaction(*args, Bq, K, None, pair_range=(2, N), want_loss=False)["logits"][2:]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([a, b]), full["logits"])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_interaction_attention_fusion(dtype):
    """InteractionRerankModel.forward with `preflmr_scores` (interaction_rerank_model.py:131-142): device-built additive
    bias over [query | context], against the stock-HF golden and the same-rounding oracle; MORES refuses it."""
    import rmr_amd
    g = _load("int_fuse_tiny")
    g0 = _load("int_tiny")
    cfg, Bq, K = g["cfg"], g["Bq"], g["K"]
    w = O.make_interaction_weights(cfg, False, seed=0)
    conf = dict(cross_encoder_num_hidden_layers=cfg.ce_layers, cross_encoder_max_position_embeddings=cfg.ce_max_pos,
                loss_fn=cfg.loss_fn, interaction_type="NORMAL", arch=arch_from_cfg(cfg, False, dtype))
    m = rmr_amd.InteractionRerankModel(conf, state_dict=w)
    t = lambda k: torch.from_numpy(g[k])
    mult = float(g["fusion_multiplier"])
    out = m(t("query_li").cuda(), t("context_li").cuda(), K - 1, t("query_mask").cuda(), t("context_mask").cuda(),
            preflmr_scores=t("preflmr_scores").cuda(), fusion_multiplier=mult, labels=g["labels_list"])
    torch.cuda.synchronize()
    gold = torch.from_numpy(g["logits"])
    d = (out.logits.cpu() - gold).abs().max().item()
    with torch.no_grad(), O.device_rounding(torch.bfloat16 if dtype == "bf16" else torch.float16) as mm:
        emu = O.interaction_forward(cfg, w, t("query_li"), t("context_li"), t("query_mask"), t("context_mask"), K,
                                    g["labels_list"], False, mm=mm, preflmr_scores=t("preflmr_scores"), fusion_multiplier=mult)
    demu = (out.logits.cpu() - emu.logits).abs().max().item()
    effect = (gold - torch.from_numpy(g0["logits"])).abs().max().item()
    print(f"[int_fuse_tiny/{dtype}] |dlogit| vs fp32 golden {d:.2e}, vs same-rounding oracle {demu:.2e}; fusion moves the logits by {effect:.2e}")
    assert effect > 5e-3 and d < (3e-4 if dtype == "fp16" else 1.5e-3) and demu < 3e-4
    assert abs(out.loss.item() - float(g["loss"])) < 2e-3
    with pytest.raises(AssertionError):                      # scores must be [N, Lc, Lq]
        m(t("query_li").cuda(), t("context_li").cuda(), K - 1, t("query_mask").cuda(), t("context_mask").cuda(),
          preflmr_scores=torch.zeros(2, 2, 2).cuda())
    gm = _load("mores_tiny")
    wm = O.make_interaction_weights(gm["cfg"], True, seed=0)
    mm_ = rmr_amd.InteractionRerankModel(dict(cross_encoder_num_hidden_layers=gm["cfg"].ce_layers,
                                              cross_encoder_max_position_embeddings=gm["cfg"].ce_max_pos,
                                              loss_fn=gm["cfg"].loss_fn, interaction_type="MORES",
                                              arch=arch_from_cfg(gm["cfg"], False, dtype)), state_dict=wm)
    tm = lambda k: torch.from_numpy(gm[k]).cuda()
    with pytest.raises(NotImplementedError):                 # mores_model.py:72-73
        mm_(tm("query_li"), tm("context_li"), gm["K"] - 1, tm("query_mask"), tm("context_mask"),
            preflmr_scores=torch.zeros(gm["Bq"] * gm["K"], gm["Lc"], gm["Lq"]).cuda())
