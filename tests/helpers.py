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
