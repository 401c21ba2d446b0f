"""Golden-vector generator (runs ONLY in the build container; never on the GPU box).

The reference's own modules cannot be imported here (missing `pytorch_lightning`,
`easydict`, `peft`, `faiss`, `colbert`, `tkinter`; pinned `transformers==4.38.2` absent —
SURVEY.md §8c: ordinary ImportErrors, nothing permission-denied) and the reference
ships no golden vectors for this path.  The arithmetic of the path lives in stock
HuggingFace `BertModel` / `BertEncoder`; this script therefore assembles those stock
modules (installed transformers, eager attention) in the order the reference composes
them (`/root/reference/src/models/rerank/rerank_model.py:333-479,523-591`,
`utils.py:73-108,228-254`, `modeling_flmr.py:603-664,1616-1688`), loads the seeded
synthetic weights by the reference's state_dict key names, runs fp32 on CPU and
writes inputs + expected outputs to `tests/golden/*.npz`.

It also cross-checks `oracle/rerank_oracle.py` against the HF assembly (max-abs diff
printed and stored) — that is the oracle's pin.

Usage:  python tests/golden/make_golden.py [--which tiny,tiny_mm,c1,...]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rerank_oracle as O  # noqa: E402

from transformers import BertConfig, BertModel  # noqa: E402
from transformers.models.bert.modeling_bert import BertEncoder  # noqa: E402


def hf_cfg(hidden, layers, heads, inter, max_pos, vocab, eps, **kw):
    c = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                   num_attention_heads=heads, intermediate_size=inter,
                   max_position_embeddings=max_pos, type_vocab_size=2, hidden_act="gelu",
                   hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                   layer_norm_eps=eps, **kw)
    c._attn_implementation = "eager"
    return c


def load_prefixed(module, w, prefix):
    sd = module.state_dict()
    new = {}
    for k in sd:
        full = prefix + k
        if full in w:
            new[k] = w[full].clone()
        else:
            new[k] = sd[k]          # unused tensors (pooler, CE word embeddings) keep their init
    module.load_state_dict(new)
    module.eval()


class HFAssembly:
    """Stock-HF restatement of the module graph the reference builds."""

    def __init__(self, cfg: O.OracleConfig, w, vision: bool):
        self.cfg, self.w, self.vision = cfg, w, vision
        self.text = BertModel(hf_cfg(cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate, cfg.max_pos,
                                     cfg.vocab_size, cfg.ln_eps), add_pooling_layer=True)
        load_prefixed(self.text, w, "context_text_encoder.bert_model.")
        self.ce = BertModel(hf_cfg(cfg.ce_hidden, cfg.ce_layers, cfg.ce_heads, cfg.ce_intermediate,
                                   cfg.ce_max_pos, cfg.vocab_size, cfg.ln_eps), add_pooling_layer=True)
        load_prefixed(self.ce, w, "reranker.bert_model.")
        if vision:
            mc = hf_cfg(cfg.hidden, cfg.map_layers, cfg.heads, cfg.intermediate, cfg.max_pos,
                        cfg.vocab_size, cfg.ln_eps, is_decoder=True, add_cross_attention=True)
            self.mapnet = BertEncoder(mc)
            load_prefixed(self.mapnet, w, "transformer_mapping_network.")

    def lin(self, x, name):
        b = self.w.get(name + ".bias")
        return torch.nn.functional.linear(x, self.w[name + ".weight"], b)

    @torch.no_grad()
    def forward(self, ids, am, tt, Bq, K, img_cls=None, img_patches=None, labels=None):
        cfg = self.cfg
        N = Bq * K
        hs = self.text(input_ids=ids, attention_mask=am, token_type_ids=tt).last_hidden_state
        text = self.lin(hs, "context_text_encoder_linear")
        mask = (ids != 0).float()
        text = text * mask.unsqueeze(2)
        Q = text
        if self.vision:
            c = img_cls.repeat_interleave(K, 0)
            pt = img_patches.repeat_interleave(K, 0)
            x = self.lin(torch.tanh(self.lin(c, "context_vision_projection.model.0")),
                         "context_vision_projection.model.2").view(N, -1, cfg.li_dim)
            t = self.lin(pt, "transformer_mapping_input_linear")
            enc = hs[:, : cfg.cross_attn_len]
            enc_mask = torch.zeros(N, 1, 1, enc.shape[1])      # inverted all-ones mask
            t = self.mapnet(t, encoder_hidden_states=enc, encoder_attention_mask=enc_mask).last_hidden_state
            t = self.lin(t, "transformer_mapping_output_linear")
            Q = torch.cat([text, x, t], dim=1)
        Q = torch.nn.functional.normalize(Q, p=2, dim=2)
        x = self.lin(Q, "cross_encoder_input_mapping")
        P = x.shape[1] - mask.shape[1]
        m = torch.cat([mask, torch.ones(N, P)], 1) if P else mask
        h = self.ce(inputs_embeds=x, attention_mask=m).last_hidden_state
        cls = h[:, 0]
        l1, l2 = self.lin(cls, "reranker.classifier1"), self.lin(cls, "reranker.classifier2")
        logits, lab = O.prepare_logits_labels(cfg.loss_fn, l1, l2, Bq, K - 1, labels)
        loss = O.loss_value(cfg.loss_fn, cfg.pos_weight, logits, lab)
        if cfg.loss_fn == "2H_BCE":
            logits = logits[:, 1].unsqueeze(1)
        return loss, logits, hs, Q


class HFInteraction:
    """Stock-HF restatement of InteractionRerankModel (interaction_rerank_model.py:96-166): NORMAL = BertModel on
    inputs_embeds; MORES = HF BertLayer(is_decoder, add_cross_attention) sub-modules called in the order of
    MORES_BertLayer.forward (mores_model.py:21-57): crossattention -> attention -> feed-forward."""

    def __init__(self, cfg: O.OracleConfig, w, mores: bool):
        from transformers.models.bert.modeling_bert import BertLayer
        self.cfg, self.w, self.mores = cfg, w, mores
        if mores:
            c = hf_cfg(cfg.ce_hidden, cfg.ce_layers, cfg.ce_heads, cfg.ce_intermediate, cfg.ce_max_pos, cfg.vocab_size,
                       cfg.ln_eps, is_decoder=True, add_cross_attention=True)
            self.layers = []
            for i in range(cfg.ce_layers):
                L = BertLayer(c, layer_idx=i)
                load_prefixed(L, w, f"reranker.interaction_module.{i}.")
                self.layers.append(L)
        else:
            self.ce = BertModel(hf_cfg(cfg.ce_hidden, cfg.ce_layers, cfg.ce_heads, cfg.ce_intermediate, cfg.ce_max_pos,
                                       cfg.vocab_size, cfg.ln_eps), add_pooling_layer=True)
            load_prefixed(self.ce, w, "reranker.bert_model.")

    def lin(self, x, name):
        return torch.nn.functional.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))

    @torch.no_grad()
    def forward(self, q, c, qm, cm, K, labels, scores=None, mult=1.0):
        cfg = self.cfg
        Bq = q.shape[0]
        qq, qmm = q.repeat_interleave(K, 0), qm.repeat_interleave(K, 0)
        if self.mores:
            h = self.lin(qq, "cross_encoder_input_mapping")
            doc = self.lin(c, "cross_encoder_input_mapping")
            fmin = torch.finfo(torch.float32).min
            qb = (1.0 - qmm)[:, None, None, :] * fmin
            cb = (1.0 - cm)[:, None, None, :] * fmin
            for L in self.layers:
                a, _ = L.crossattention(h, None, doc, cb)
                b, _ = L.attention(a, qb)
                h = L.feed_forward_chunk(b)
            cls = h[:, 0]
        else:
            x = self.lin(torch.cat((qq, c), 1), "cross_encoder_input_mapping")
            m01 = torch.cat((qmm, cm), 1)
            if scores is None:
                cls = self.ce(inputs_embeds=x, attention_mask=m01).last_hidden_state[:, 0]
            else:     # attention fusion (interaction_rerank_model.py:131-142): stock BertEncoder on the additive 4-D mask
                N, Lq, Lc = x.shape[0], q.shape[1], c.shape[1]
                ur, bl = torch.softmax(scores.permute(0, 2, 1), -1), torch.softmax(scores, -1)
                adj = torch.cat([torch.cat([torch.zeros(N, Lq, Lq), ur], 2), torch.cat([bl, torch.zeros(N, Lc, Lc)], 2)], 1) * mult
                ext = (1.0 - m01)[:, None, None, :] * torch.finfo(torch.float32).min + adj[:, None]
                cls = self.ce.encoder(self.ce.embeddings(inputs_embeds=x), attention_mask=ext).last_hidden_state[:, 0]
        l1, l2 = self.lin(cls, "reranker.classifier1"), self.lin(cls, "reranker.classifier2")
        logits, lab = O.prepare_logits_labels(cfg.loss_fn, l1, l2, Bq, K - 1, labels)
        loss = O.loss_value(cfg.loss_fn, cfg.pos_weight, logits, lab)
        if cfg.loss_fn == "2H_BCE":
            logits = logits[:, 1].unsqueeze(1)
        return loss, logits


def run_rerankmodel_case(outdir, name="rm_tiny", fusion=False):
    """RerankModel.forward (ids signature) with instruction masking and the [query|image|context] reorder; `fusion`
    adds the PreFLMR attention-fusion bias (rerank_model.py:276-319) from seeded retriever scores."""
    kw = dict(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=512, max_pos=64, ce_hidden=128, ce_heads=2,
              ce_intermediate=512, ce_layers=2, ce_max_pos=128, li_dim=64, vision_hidden=128, prefix_len=4, n_patches=9,
              cross_attn_len=32, pos_weight=2.0)
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = "2H_BCE"                      # the one reference config using RerankModel is the 2-head variant
    Bq, K, S, ql, instr = 2, 3, 64, 8, 1999
    w = O.make_weights(cfg, seed=0, vision=True)
    rng = np.random.Generator(np.random.PCG64(7))
    q_ids = torch.from_numpy(rng.integers(1000, 1990, size=(Bq, ql)))
    q_ids[:, 0] = 101
    q_ids[0, 5] = instr                          # query 0 has an instruction separator at position 5; query 1 has none
    q_am = torch.ones(Bq, ql, dtype=torch.int64)
    c_ids, c_am, _ = O.make_pair_batch(cfg, Bq, K, S, seed=3, regime="realistic")
    img = O.make_image_feats(cfg, Bq)
    hf = HFAssembly(cfg, w, True)
    # HF restatement of the same graph
    with torch.no_grad():
        N = Bq * K
        jq, jm = q_ids.repeat_interleave(K, 0), q_am.repeat_interleave(K, 0)
        j_ids = torch.cat([jq, c_ids[:, 2:2 - ql]], 1)
        j_am = torch.cat([jm, c_am[:, 2:2 - ql]], 1)
        hs = hf.text(input_ids=j_ids, attention_mask=j_am).last_hidden_state
        mask = O.instruction_query_mask(j_ids, instr)
        text = hf.lin(hs, "context_text_encoder_linear") * mask.unsqueeze(2)
        cc, pt = img[0].repeat_interleave(K, 0), img[1].repeat_interleave(K, 0)
        x = hf.lin(torch.tanh(hf.lin(cc, "context_vision_projection.model.0")),
                   "context_vision_projection.model.2").view(N, -1, cfg.li_dim)
        t = hf.lin(pt, "transformer_mapping_input_linear")
        enc = hs[:, : cfg.cross_attn_len]
        t = hf.mapnet(t, encoder_hidden_states=enc, encoder_attention_mask=torch.zeros(N, 1, 1, enc.shape[1])).last_hidden_state
        t = hf.lin(t, "transformer_mapping_output_linear")
        Q = torch.nn.functional.normalize(torch.cat([text, x, t], 1), p=2, dim=2)
        xin = hf.lin(Q, "cross_encoder_input_mapping")
        m = torch.cat([mask, torch.ones(N, xin.shape[1] - S)], 1)
        xin = torch.cat((xin[:, :ql], xin[:, S:], xin[:, ql:S]), 1)
        m = torch.cat((m[:, :ql], m[:, S:], m[:, ql:S]), 1)
        scores, mult = None, 1.0
        if fusion:
            gs = torch.Generator().manual_seed(99)
            scores = 3.0 * torch.randn(N, S, ql + xin.shape[1] - S, generator=gs)      # retriever scores_raw stand-in
            mult = 20.0
            adj = O.fusion_adjacency(scores, ql, xin.shape[1] - S, S, mult)
            ext = (1.0 - m)[:, None, None, :] * torch.finfo(torch.float32).min + adj[:, None]
            emb = hf.ce.embeddings(inputs_embeds=xin)
            cls = hf.ce.encoder(emb, attention_mask=ext).last_hidden_state[:, 0]       # stock BertEncoder, additive 4-D mask
        else:
            cls = hf.ce(inputs_embeds=xin, attention_mask=m).last_hidden_state[:, 0]
        l12 = torch.cat((hf.lin(cls, "reranker.classifier1"), hf.lin(cls, "reranker.classifier2")), 1)
        loss_hf = torch.nn.functional.cross_entropy(l12, l12, weight=torch.tensor([1.0, cfg.pos_weight]))
        logits_hf = l12[:, 1:2]
        out = O.rerank_model_forward(cfg, w, q_ids, q_am, c_ids, c_am, K, img[0], img[1], instr, preflmr_scores=scores,
                                     fusion_multiplier=mult)
    d_logit, d_loss = (out.logits - logits_hf).abs().max().item(), (out.loss - loss_hf).abs().item()
    print(f"[{name}] oracle-vs-HF: logits {d_logit:.3e} loss {d_loss:.3e}")
    np.savez_compressed(os.path.join(outdir, f"{name}.npz"), cfg_json=np.array(repr(kw)), Bq=Bq, K=K, S=S, ql=ql,
                        instruction_token_id=instr, query_input_ids=q_ids.numpy(), query_attention_mask=q_am.numpy(),
                        context_input_ids=c_ids.numpy(), context_attention_mask=c_am.numpy(),
                        image_cls=img[0].numpy(), image_patches=img[1].numpy(), logits=logits_hf.numpy(),
                        loss=np.array(loss_hf.item(), dtype=np.float32), oracle_vs_hf=np.array([d_logit, d_loss]),
                        fusion=np.array(int(fusion)), fusion_multiplier=np.array(mult, dtype=np.float32),
                        preflmr_scores=(scores.numpy() if fusion else np.zeros(0, dtype=np.float32)))


VIT_CASES = {
    # name: (cfg kwargs, B)   — the CLIP vision tower the rerankers call once per query (rerank_model.py:408-426)
    "vit_tiny": (dict(vision_hidden=128, n_patches=16, vit_layers=3, vit_heads=2, vit_intermediate=256,
                      vit_image_size=64, vit_patch_size=16), 3),
    "vit_p14": (dict(vision_hidden=128, n_patches=16, vit_layers=2, vit_heads=2, vit_intermediate=256,
                     vit_image_size=56, vit_patch_size=14), 2),           # 3*14*14 = 588: K not a multiple of 64
    "vit_b32": (dict(), 2),                                                # openai/clip-vit-base-patch32 shape
}


def run_vit_case(name, outdir):
    """Stock HF CLIPVisionModel with the seeded weights; pixel values are regenerated from the seed by the tests."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    kw, B = VIT_CASES[name]
    cfg = O.OracleConfig(**kw)
    w = O.make_vit_weights(cfg, seed=5)
    hc = CLIPVisionConfig(hidden_size=cfg.vision_hidden, intermediate_size=cfg.vit_intermediate,
                          num_hidden_layers=cfg.vit_layers, num_attention_heads=cfg.vit_heads,
                          image_size=cfg.vit_image_size, patch_size=cfg.vit_patch_size, hidden_act="quick_gelu",
                          layer_norm_eps=1e-5, attn_implementation="eager")
    m = CLIPVisionModel(hc).eval()
    sd = m.state_dict()
    pref = "vision_model." if next(iter(sd)).startswith("vision_model.") else ""   # transformers 4.x vs 5.x key layout
    new = {k: w.get(O.VIT_PREFIX + "." + k[len(pref):], t) for k, t in sd.items()}
    assert sum(O.VIT_PREFIX + "." + k[len(pref):] in w for k in sd) == len(w)
    m.load_state_dict(new)
    px = O.make_pixel_values(cfg, B, seed=2022)
    with torch.no_grad():
        out = m(pixel_values=px, output_hidden_states=True)
        cls_hf, pat_hf = out.last_hidden_state[:, 0], out.hidden_states[-2][:, 1:]
        c, p_ = O.clip_vision_forward(cfg, w, px)
    d = [(c - cls_hf).abs().max().item(), (p_ - pat_hf).abs().max().item()]
    print(f"[{name}] oracle-vs-HF: cls {d[0]:.3e} patches {d[1]:.3e}")
    np.savez_compressed(os.path.join(outdir, f"{name}.npz"), cfg_json=np.array(repr(kw)), B=B, weight_seed=5,
                        pixel_seed=2022, pixel_checksum=np.array(px.double().sum().item()),
                        image_cls=cls_hf.numpy(), image_patches=pat_hf.numpy(), oracle_vs_hf=np.array(d))


INTERACTION_CASES = {
    # name: (cfg kwargs, Bq, K, Lq, Lc, mores, loss)
    "int_tiny": (dict(ce_hidden=128, ce_heads=2, ce_intermediate=512, ce_layers=2, ce_max_pos=128, li_dim=64),
                 2, 3, 9, 40, False, "BCE"),
    "mores_tiny": (dict(ce_hidden=128, ce_heads=2, ce_intermediate=512, ce_layers=2, ce_max_pos=128, li_dim=64),
                   2, 3, 9, 40, True, "negative_sampling"),
    "int_fuse_tiny": (dict(ce_hidden=128, ce_heads=2, ce_intermediate=512, ce_layers=2, ce_max_pos=128, li_dim=64),
                      2, 3, 9, 40, False, "BCE"),                             # + PreFLMR attention fusion
    "int_base": (dict(ce_layers=3), 1, 6, 113, 512, False, "BCE"),          # ModPreFLMR-BERT shape, K cut to 6
    "mores_base": (dict(ce_layers=5), 1, 6, 113, 512, True, "BCE"),         # ModPreFLMR-IB shape
}


def run_interaction_case(name, outdir):
    kw, Bq, K, Lq, Lc, mores, loss = INTERACTION_CASES[name]
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = loss
    w = O.make_interaction_weights(cfg, mores, seed=0)
    q, c, qm, cm = O.make_interaction_inputs(cfg, Bq, K, Lq, Lc)
    labels = None
    if loss != "negative_sampling":
        rng = np.random.Generator(np.random.PCG64(5))
        labels = [float(x) for x in (rng.random(Bq * K) < 0.3)]
    scores, mult = None, 1.0
    if "fuse" in name:
        scores = 3.0 * torch.randn(Bq * K, Lc, Lq, generator=torch.Generator().manual_seed(77))
        mult = 20.0
    loss_hf, logits_hf = HFInteraction(cfg, w, mores).forward(q, c, qm, cm, K, labels, scores, mult)
    out = O.interaction_forward(cfg, w, q, c, qm, cm, K, labels, mores, preflmr_scores=scores, fusion_multiplier=mult)
    d_logit = (out.logits - logits_hf).abs().max().item()
    d_loss = (out.loss - loss_hf).abs().item()
    print(f"[{name}] oracle-vs-HF: logits {d_logit:.3e} loss {d_loss:.3e}")
    order = [O.rank_descending_stable(r) for r in logits_hf.view(Bq, -1).tolist()]
    np.savez_compressed(os.path.join(outdir, f"{name}.npz"), cfg_json=np.array(repr(kw)), Bq=Bq, K=K, Lq=Lq, Lc=Lc,
                        mores=mores, loss_fn=np.array(loss), query_li=q.numpy(), context_li=c.numpy(),
                        query_mask=qm.numpy(), context_mask=cm.numpy(),
                        labels=np.array(labels if labels is not None else [], dtype=np.float32),
                        logits=logits_hf.numpy(), loss=np.array(loss_hf.item(), dtype=np.float32),
                        order=np.array(order, dtype=np.int32), oracle_vs_hf=np.array([d_logit, d_loss]),
                        preflmr_scores=(scores.numpy() if scores is not None else np.zeros(0, dtype=np.float32)),
                        fusion_multiplier=np.array(mult, dtype=np.float32))


CASES = {
    # name: (cfg kwargs, Bq, K, S, vision, regime, loss)
    "tiny": (dict(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=512, max_pos=64,
                  ce_hidden=128, ce_heads=2, ce_intermediate=512, ce_layers=1, ce_max_pos=128,
                  li_dim=64), 2, 3, 64, False, "realistic", "BCE"),
    "tiny_mm": (dict(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=512, max_pos=64,
                     ce_hidden=128, ce_heads=2, ce_intermediate=512, ce_layers=2, ce_max_pos=128,
                     li_dim=64, vision_hidden=128, prefix_len=4, n_patches=9, cross_attn_len=32),
                2, 3, 64, True, "realistic", "negative_sampling"),
    "tiny_2h": (dict(vocab_size=2000, hidden=128, layers=1, heads=2, intermediate=512, max_pos=64,
                     ce_hidden=128, ce_heads=2, ce_intermediate=512, ce_layers=1, ce_max_pos=128,
                     li_dim=64, pos_weight=3.0), 2, 4, 64, False, "realistic", "2H_BCE"),
    "c1": (dict(), 2, 5, 128, False, "realistic", "BCE"),                 # BASELINE configs[0]
    "c2": (dict(), 1, 20, 256, False, "realistic", "BCE"),                # BASELINE configs[1]
    "c3s": (dict(), 1, 4, 512, True, "realistic", "negative_sampling"),   # configs[2] shape, K cut to 4
}


def run_case(name, outdir):
    kw, Bq, K, S, vision, regime, loss = CASES[name]
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = loss
    w = O.make_weights(cfg, seed=0, vision=vision)
    ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=2022, regime=regime)
    img = O.make_image_feats(cfg, Bq) if vision else (None, None)
    labels = None
    if loss != "negative_sampling":
        rng = np.random.Generator(np.random.PCG64(5))
        labels = [float(x) for x in (rng.random(Bq * K) < 0.3)]
    hf = HFAssembly(cfg, w, vision)
    loss_hf, logits_hf, hs_hf, Q_hf = hf.forward(ids, am, tt, Bq, K, img[0], img[1], labels)
    out = O.full_context_forward(cfg, w, ids, am, tt, Bq, K, img[0], img[1], labels, want_taps=True)
    d_logit = (out.logits - logits_hf).abs().max().item()
    d_loss = (out.loss - loss_hf).abs().item()
    d_hs = (out.taps[f"text_layer_{cfg.layers - 1}"] - hs_hf).abs().max().item()
    d_q = (out.taps["late_interaction"] - Q_hf).abs().max().item()
    print(f"[{name}] oracle-vs-HF: logits {d_logit:.3e} loss {d_loss:.3e} text_hidden {d_hs:.3e} LI {d_q:.3e}")
    # weight checksums guard against RNG drift between torch builds
    wsum = np.array([float(w[k].double().sum()) for k in sorted(w)][:64])
    order = [O.rank_descending_stable(r) for r in
             logits_hf.view(Bq, -1).tolist()] if logits_hf.numel() == Bq * K else []
    np.savez_compressed(
        os.path.join(outdir, f"{name}.npz"),
        cfg_json=np.array(repr(kw)), Bq=Bq, K=K, S=S, vision=vision, loss_fn=np.array(loss),
        pos_weight=np.array(cfg.pos_weight if cfg.pos_weight is not None else np.nan),
        input_ids=ids.numpy(), attention_mask=am.numpy(), token_type_ids=tt.numpy(),
        image_cls=(img[0].numpy() if vision else np.zeros(0, np.float32)),
        image_patches=(img[1].numpy() if vision else np.zeros(0, np.float32)),
        labels=np.array(labels if labels is not None else [], dtype=np.float32),
        logits=logits_hf.numpy(), loss=np.array(loss_hf.item(), dtype=np.float32),
        order=np.array(order, dtype=np.int32),
        text_hidden_cls=hs_hf[:, 0].numpy(),                  # CLS row of the 12-layer encoder
        text_hidden_absmean=np.array(hs_hf.abs().mean().item()),
        late_interaction_row0=Q_hf[:, 0].numpy(), weight_sums=wsum,
        oracle_vs_hf=np.array([d_logit, d_loss, d_hs, d_q]),
    )



# ----------------------------------------------------------------------------------------------------------------------
# bf16-autocast goldens: the reference's ACTUAL arithmetic (`precision: 'bf16'` = Lightning bf16-mixed,
# configs/Rerank/OKVQA/Encoder/monoPreFLMR-B_pointwise.jsonnet:186,233): fp32 parameters, torch.autocast(bfloat16) around
# the forward.  This container has no GPU, so the CUDA autocast policy is emulated on the CPU: torch.autocast("cpu",
# bfloat16) already runs linear / matmul / bmm in bf16 (fp32 accumulate, bf16 result); what the CPU policy lacks is the
# CUDA policy's fp32 list, of which this path touches softmax and layer_norm — both are forced to fp32 here, as
# `torch/csrc/autograd/autocast_mode.cpp` (CUDA section: softmax, layer_norm, ... under KERNEL(..., fp32)) does on a GPU.
import contextlib  # noqa: E402


@contextlib.contextmanager
def cuda_autocast_emulation():
    import torch.nn.functional as F
    real_softmax, real_ln = F.softmax, F.layer_norm

    def softmax32(input, dim=None, _stacklevel=3, dtype=None):
        with torch.autocast("cpu", enabled=False):
            return real_softmax(input.float(), dim=dim)

    def ln32(input, normalized_shape, weight=None, bias=None, eps=1e-5):
        with torch.autocast("cpu", enabled=False):
            return real_ln(input.float(), normalized_shape, None if weight is None else weight.float(),
                           None if bias is None else bias.float(), eps)

    F.softmax, F.layer_norm = softmax32, ln32
    keep_t = torch.softmax
    torch.softmax = lambda x, dim=-1, dtype=None: softmax32(x, dim)
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            yield
    finally:
        F.softmax, F.layer_norm, torch.softmax = real_softmax, real_ln, keep_t


def _labels_for(loss, n):
    if loss == "negative_sampling":
        return None
    rng = np.random.Generator(np.random.PCG64(5))
    return [float(x) for x in (rng.random(n) < 0.3)]


def run_autocast_goldens(outdir, names):
    """logits / loss of the stock-HF assembly under the emulated CUDA bf16 autocast, for the cases of CASES and
    INTERACTION_CASES named in `names` -> autocast.npz (keys `<case>.logits`, `<case>.loss`).  The gate of the GPU tests
    for the bf16 mode is |device - fp32| <= max(1e-3, |autocast - fp32|) per case."""
    path = os.path.join(outdir, "autocast.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in names:
        if name in CASES:
            kw, Bq, K, S, vision, regime, loss = CASES[name]
            cfg = O.OracleConfig(**kw)
            cfg.loss_fn = loss
            w = O.make_weights(cfg, seed=0, vision=vision)
            ids, am, tt = O.make_pair_batch(cfg, Bq, K, S, seed=2022, regime=regime)
            img = O.make_image_feats(cfg, Bq) if vision else (None, None)
            labels = _labels_for(loss, Bq * K)
            hf = HFAssembly(cfg, w, vision)
            _, ref, _, _ = hf.forward(ids, am, tt, Bq, K, img[0], img[1], labels)
            with cuda_autocast_emulation():
                l_ac, lg_ac, _, _ = hf.forward(ids, am, tt, Bq, K, img[0], img[1], labels)
        else:
            kw, Bq, K, Lq, Lc, mores, loss = INTERACTION_CASES[name]
            cfg = O.OracleConfig(**kw)
            cfg.loss_fn = loss
            w = O.make_interaction_weights(cfg, mores, seed=0)
            q, c, qm, cm = O.make_interaction_inputs(cfg, Bq, K, Lq, Lc)
            labels = _labels_for(loss, Bq * K)
            hf = HFInteraction(cfg, w, mores)
            _, ref = hf.forward(q, c, qm, cm, K, labels)
            with cuda_autocast_emulation():
                l_ac, lg_ac = hf.forward(q, c, qm, cm, K, labels)
        lg = lg_ac.float().reshape(-1)
        d = (lg - ref.reshape(-1)).abs().max().item()
        print(f"[autocast {name}] |autocast - fp32| max {d:.3e}  (logits dtype under autocast: {lg_ac.dtype})")
        out[f"{name}.logits"] = lg.numpy()
        out[f"{name}.loss"] = np.array(float(l_ac), dtype=np.float32)
        out[f"{name}.drift_vs_fp32"] = np.array(d, dtype=np.float32)
    np.savez_compressed(path, **out)


def _hf_logits_chunked(hf, ids, am, tt, K, img, chunk=100):
    """Pointwise logits of ONE query's K candidates in chunks (pairs are independent: rerank_model.py:541-555)."""
    outs = []
    for b in range(0, K, chunk):
        e = min(K, b + chunk)
        _, lg, _, _ = hf.forward(ids[b:e], am[b:e], tt[b:e], 1, e - b, img[0], img[1], None)
        outs.append(lg.float().reshape(-1))
    return torch.cat(outs)


FULLSIZE = {
    # name: (cfg kwargs, vision, S, pool per query)        BASELINE configs[2] / configs[4] / monoPreFLMR-L shapes
    "c3_full": (dict(), True, 512, 100),
    "c3_sep": (dict(), True, 512, 300),
    "c5_full": (dict(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16,
                     ce_intermediate=4096), False, 512, 200),
    "l_shape": (dict(vision_hidden=1024, n_patches=256, ce_max_pos=900), True, 512, 8),   # monoPreFLMR-L_pointwise.jsonnet:117
    # configs[4]'s ranking fixture: bert-large, K = 200 chosen from a pool of 300, widened weights as c3_sep
    "c5_sep": (dict(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16,
                    ce_intermediate=4096), False, 512, 300),
}
_LARGE = FULLSIZE["c5_sep"]
# the same construction at lower gains, and at gain 2.5 with the WIDEST gap a pool of 300 allows for K = 200
# (the 100 candidates behind fp32 rank 5 are left out), so that every fixture carries what the reference's own bf16-autocast
# arithmetic does on it.  `gap_rel` = designed gap as a fraction of the pool's logit standard deviation (c5_sep: 0.12 / 0.28).
FULLSIZE.update({"c5_sep_g20": _LARGE, "c5_sep_g15": _LARGE, "c5_sep_wide": _LARGE})
SEP = {"c3_sep": dict(K=100, gap=0.08), "c5_sep": dict(K=200, gap=float(os.environ.get("RR_C5_SEP_GAP", "0.12"))),
       "c5_sep_g20": dict(K=200, gap_rel=0.4, gain=2.0), "c5_sep_g15": dict(K=200, gap_rel=0.4, gain=1.5),
       "c5_sep_wide": dict(K=200, widest=True, pool_of="c5_sep")}


def run_fullsize_case(name, outdir):
    """Full-size goldens (fp32 stock-HF logits + the bf16-autocast logits of the same lists).
    c3_full: K = 100, S = 512, P = 81 (BASELINE configs[2]), one query, the list as drawn.
    c3_sep: the same shape with the Linear matrices widened (make_weights gain 2.5: a 0.02-std random network scores all
      candidates of a query within +-0.02, i.e. at the bf16 noise level; widened, the logit spread is ~0.2) and, for TWO
      queries, a list of 100 chosen from a pool of 300 seeded candidates so that the fp32 logits leave a gap of >= 0.08
      between rank 5 and rank 6 (pairs are scored independently: choosing the list changes no logit).  The Recall@5 /
      top-5 parity test is then a statement about the kernels, not about ties: query 0's only positive sits at fp32 rank
      5, query 1's at rank 6, so Recall@5 must come out 1 and 0.
    c5_full: bert-large text-only, K = 200, S = 512 (BASELINE configs[4] shape) as drawn.
    l_shape: monoPreFLMR-L geometry (ViT-L/14 features: 1024-d, 256 patches, P = 288, T = 800, position table 900), K = 8."""
    kw, vision, S, pool = FULLSIZE[name]
    cfg = O.OracleConfig(**kw)
    cfg.loss_fn = "BCE"
    gain = SEP[name].get("gain", 2.5) if name in SEP else 1.0
    w = O.make_weights(cfg, seed=0, vision=vision, gain=gain)
    hf = HFAssembly(cfg, w, vision)
    nq = 2 if name in SEP else 1
    rec = dict(cfg_json=np.array(repr(kw)), S=S, vision=vision, pool=pool, nq=nq, gain=np.array(gain, dtype=np.float32))
    for qi in range(nq):
        seed = 2022 + 31 * qi
        ids, am, tt = O.make_pair_batch(cfg, 1, pool, S, seed=seed, regime="realistic")
        # one query: the query span (tokens 1..32) is the same text for all its candidates
        ids[:, 1:33] = ids[0, 1:33]
        img = O.make_image_feats(cfg, 1, seed=seed) if vision else (None, None)
        import time
        t0 = time.time()
        cache = os.environ.get("RR_GOLDEN_POOL_CACHE")       # re-select a list (another gap) without the hour of CPU forwards
        pool_name = SEP.get(name, {}).get("pool_of", name)       # c5_sep_wide re-selects from c5_sep's pool: same logits
        cache_ac = os.path.join(cache, f"{pool_name}_q{qi}_pool_autocast.pt") if cache else None
        cache = os.path.join(cache, f"{pool_name}_q{qi}_pool_fp32.pt") if cache else None
        chunk = 50 if kw.get("hidden", 768) > 768 else 100
        if cache and os.path.exists(cache):
            fp32 = torch.load(cache)
        else:
            fp32 = _hf_logits_chunked(hf, ids, am, tt, pool, img, chunk=chunk)
            if cache:
                torch.save(fp32, cache)
        t1 = time.time()
        # the reference's own arithmetic (bf16 autocast) on the same pool: the yardstick of every reduced-precision mode
        if cache_ac and os.path.exists(cache_ac):
            ac = torch.load(cache_ac)
        else:
            with cuda_autocast_emulation():
                ac = _hf_logits_chunked(hf, ids, am, tt, pool, img, chunk=chunk)
            if cache_ac:
                torch.save(ac, cache_ac)
        print(f"[{name} q{qi}] pool of {pool}: fp32 {t1 - t0:.0f} s, autocast {time.time() - t1:.0f} s; "
              f"|autocast - fp32| max {(ac - fp32).abs().max():.3e}; logit std {fp32.std():.3f}")
        rec[f"q{qi}.seed"] = seed
        rec[f"q{qi}.ids_checksum"] = np.array(int(ids.sum()))
        rec[f"q{qi}.pool_logits"] = fp32.numpy()
        rec[f"q{qi}.pool_logits_autocast"] = ac.numpy()
        if name in SEP:
            Ksel = SEP[name]["K"]
            order = sorted(range(pool), key=lambda i: -fp32[i].item())
            top5 = order[:5]
            if SEP[name].get("widest"):
                j = pool - (Ksel - 5)
                gap = float(fp32[order[4]] - fp32[order[j]])
            else:
                gap = SEP[name]["gap"] if "gap" in SEP[name] else round(SEP[name]["gap_rel"] * float(fp32.std()), 3)
                j = 5
                while j < pool and fp32[order[4]] - fp32[order[j]] < gap:
                    j += 1
            rec[f"q{qi}.gap_design"] = np.array(gap, dtype=np.float32)
            rest = order[j:j + Ksel - 5]
            assert len(rest) == Ksel - 5, f"pool too small for a {gap} gap (j={j})"
            rng = np.random.Generator(np.random.PCG64(99 + qi))
            sel = np.array(top5 + rest)
            rng.shuffle(sel)
            rec[f"q{qi}.selected"] = sel.astype(np.int32)                       # pool indices, list order
            pos_rank = 5 if qi == 0 else 6                                      # 1-based fp32 rank of the only positive
            pos_pool = (top5 + rest)[pos_rank - 1]
            rec[f"q{qi}.positive_list_index"] = np.array(int(np.where(sel == pos_pool)[0][0]))
            rec[f"q{qi}.gap_5_6"] = np.array(float(fp32[order[4]] - fp32[rest[0]]), dtype=np.float32)
            print(f"   selected list: rank-5/6 gap {rec[f'q{qi}.gap_5_6']:.4f}, positive at fp32 rank {pos_rank}")
    np.savez_compressed(os.path.join(outdir, f"{name}.npz"), **rec)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default=",".join(list(CASES) + list(INTERACTION_CASES) + ["rm_tiny", "rm_fuse_tiny"] + list(VIT_CASES)))
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("RR_GOLDEN_THREADS", "8")))
    here = os.path.dirname(os.path.abspath(__file__))
    for nm in a.which.split(","):
        if nm.startswith("autocast:"):          # e.g. autocast:c1+c2+c3s+int_base+mores_base
            run_autocast_goldens(here, nm[len("autocast:"):].split("+"))
        elif nm in FULLSIZE:
            run_fullsize_case(nm, here)
        elif nm == "rm_tiny":
            run_rerankmodel_case(os.path.dirname(os.path.abspath(__file__)))
        elif nm == "rm_fuse_tiny":
            run_rerankmodel_case(os.path.dirname(os.path.abspath(__file__)), name="rm_fuse_tiny", fusion=True)
        elif nm in INTERACTION_CASES:
            run_interaction_case(nm, os.path.dirname(os.path.abspath(__file__)))
        elif nm in VIT_CASES:
            run_vit_case(nm, os.path.dirname(os.path.abspath(__file__)))
        else:
            run_case(nm, os.path.dirname(os.path.abspath(__file__)))
