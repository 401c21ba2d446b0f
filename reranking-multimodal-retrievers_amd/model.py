"""Host-side mirror of the reference's reranker plug-in interface, on top of librerank_mi355.so.

Reference interface mirrored (paths relative to /root/reference/):
  * plug-in construction `RerankerClass(reranker_config)` — src/executors/Reranker_base_executor.py:191-202
  * `FullContextRerankModel.forward(query_text_sequences, query_pixel_values, context_text_sequences,
     num_negative_examples, labels=None)` — src/models/rerank/rerank_model.py:523-591
  * return `EasyDict(loss=<0-dim tensor>, logits=<tensor>)` consumed at Reranker_base_executor.py:922-927
  * error behaviour: AssertionError on N != Bq*K / label count (rerank_model.py:527-529), ValueError for
    labels with negative_sampling (utils.py:233), NotImplementedError for unsupported variants.

Nothing here computes: tensors are only allocated and handed to the C ABI by address.  If the HIP
library is missing, or no MI355X is visible, construction raises — there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from .pair_inputs import group_pairs_by_length


class RerankOutput(dict):
    """EasyDict-style result: `.loss` (0-dim fp32 device tensor), `.logits` (fp32 device tensor)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


def _get(cfg, name, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


# bert-base-uncased / PreFLMR ViT-B architecture (configuration_flmr.py:90-122,220-236,332-350)
FLMR_DEFAULTS = dict(vocab_size=30522, hidden=768, layers=12, heads=12, intermediate=3072, max_pos=512,
                     type_vocab=2, ln_eps=1e-12, li_dim=128, vision_hidden=768, prefix_len=32, n_patches=49,
                     map_layers=1, cross_attn_len=32)
# CLIP ViT-B/32 vision tower (FLMRVisionConfig, configuration_flmr.py:90-104); vit_layers = 0 leaves the image
# features to the caller (image_feature_fn / image_features=)
VIT_DEFAULTS = dict(vit_layers=0, vit_heads=12, vit_intermediate=3072, vit_image_size=224, vit_patch_size=32)
CE_DEFAULTS = dict(ce_hidden=768, ce_heads=12, ce_intermediate=3072)   # cross_encoder_config_base = bert-base-uncased


def make_arch(reranker_config=None, **overrides) -> dict:
    """Architecture dict from a reference-style `reranker_config` (EasyDict/dict/object with
    `cross_encoder_num_hidden_layers`, `cross_encoder_max_position_embeddings`, `loss_fn`, `pos_weight`,
    ... — monoBERT_pointwise.jsonnet:111-122) plus optional `arch` overrides for the FLMR side."""
    a = dict(FLMR_DEFAULTS)
    a.update(CE_DEFAULTS)
    a.update(VIT_DEFAULTS)
    if _get(reranker_config, "vision_encoder", False):     # run the CLIP tower inside the library
        a["vit_layers"] = 12
    a.update(ce_layers=_get(reranker_config, "cross_encoder_num_hidden_layers", 1),
             ce_max_pos=_get(reranker_config, "cross_encoder_max_position_embeddings", 750),
             loss_fn=_get(reranker_config, "loss_fn", "BCE"),
             pos_weight=_get(reranker_config, "pos_weight", None),
             has_vision=1,
             compute_dtype=_get(reranker_config, "compute_dtype", "bf16"),   # "bf16" | "fp16" MFMA operands
             model_kind="full_context")
    a.update(_get(reranker_config, "arch", None) or {})
    a.update(overrides)
    return a


def weight_spec(a: dict) -> List[tuple]:
    """(name, shape, kind) of every tensor the path reads, named by the reference state_dict keys."""
    H, I, D = a["hidden"], a["intermediate"], a["li_dim"]

    def layer(p, Hh, Ii, cross):
        out = []
        for att in ["attention"] + (["crossattention"] if cross else []):
            for n in ("query", "key", "value"):
                out += [(f"{p}.{att}.self.{n}.weight", (Hh, Hh), "w"), (f"{p}.{att}.self.{n}.bias", (Hh,), "b")]
            out += [(f"{p}.{att}.output.dense.weight", (Hh, Hh), "w"), (f"{p}.{att}.output.dense.bias", (Hh,), "b"),
                    (f"{p}.{att}.output.LayerNorm.weight", (Hh,), "g"), (f"{p}.{att}.output.LayerNorm.bias", (Hh,), "b")]
        out += [(f"{p}.intermediate.dense.weight", (Ii, Hh), "w"), (f"{p}.intermediate.dense.bias", (Ii,), "b"),
                (f"{p}.output.dense.weight", (Hh, Ii), "w"), (f"{p}.output.dense.bias", (Hh,), "b"),
                (f"{p}.output.LayerNorm.weight", (Hh,), "g"), (f"{p}.output.LayerNorm.bias", (Hh,), "b")]
        return out

    s = []
    kind = a.get("model_kind", "full_context")
    if kind != "full_context":          # InteractionRerankModel: input mapping + reranker only
        Hc, Ic = a["ce_hidden"], a["ce_intermediate"]
        s += [("cross_encoder_input_mapping.weight", (Hc, D), "w"), ("cross_encoder_input_mapping.bias", (Hc,), "b")]
        if kind == "interaction":
            p = "reranker.bert_model"
            s += [(f"{p}.embeddings.position_embeddings.weight", (a["ce_max_pos"], Hc), "e"),
                  (f"{p}.embeddings.token_type_embeddings.weight", (a["type_vocab"], Hc), "e"),
                  (f"{p}.embeddings.LayerNorm.weight", (Hc,), "g"), (f"{p}.embeddings.LayerNorm.bias", (Hc,), "b")]
            for i in range(a["ce_layers"]):
                s += layer(f"{p}.encoder.layer.{i}", Hc, Ic, False)
        else:
            for i in range(a["ce_layers"]):
                s += layer(f"reranker.interaction_module.{i}", Hc, Ic, True)
        s += [("reranker.classifier1.weight", (1, Hc), "w"), ("reranker.classifier1.bias", (1,), "b"),
              ("reranker.classifier2.weight", (1, Hc), "w"), ("reranker.classifier2.bias", (1,), "b")]
        return s
    p = "context_text_encoder.bert_model"
    s += [(f"{p}.embeddings.word_embeddings.weight", (a["vocab_size"], H), "e"),
          (f"{p}.embeddings.position_embeddings.weight", (a["max_pos"], H), "e"),
          (f"{p}.embeddings.token_type_embeddings.weight", (a["type_vocab"], H), "e"),
          (f"{p}.embeddings.LayerNorm.weight", (H,), "g"), (f"{p}.embeddings.LayerNorm.bias", (H,), "b")]
    for i in range(a["layers"]):
        s += layer(f"{p}.encoder.layer.{i}", H, I, False)
    s += [("context_text_encoder_linear.weight", (D, H), "w")]
    if a["has_vision"]:
        Vh, PL = a["vision_hidden"], a["prefix_len"]
        s += [("context_vision_projection.model.0.weight", (D * PL // 2, Vh), "w"),
              ("context_vision_projection.model.0.bias", (D * PL // 2,), "b"),
              ("context_vision_projection.model.2.weight", (D * PL, D * PL // 2), "w"),
              ("context_vision_projection.model.2.bias", (D * PL,), "b"),
              ("transformer_mapping_input_linear.weight", (H, Vh), "w"),
              ("transformer_mapping_input_linear.bias", (H,), "b")]
        for i in range(a["map_layers"]):
            s += layer(f"transformer_mapping_network.layer.{i}", H, I, True)
        s += [("transformer_mapping_output_linear.weight", (D, H), "w"),
              ("transformer_mapping_output_linear.bias", (D,), "b")]
    if a.get("vit_layers", 0) > 0:
        s += vit_weight_spec(a)
    Hc, Ic = a["ce_hidden"], a["ce_intermediate"]
    s += [("cross_encoder_input_mapping.weight", (Hc, D), "w"), ("cross_encoder_input_mapping.bias", (Hc,), "b")]
    p = "reranker.bert_model"
    s += [(f"{p}.embeddings.position_embeddings.weight", (a["ce_max_pos"], Hc), "e"),
          (f"{p}.embeddings.token_type_embeddings.weight", (a["type_vocab"], Hc), "e"),
          (f"{p}.embeddings.LayerNorm.weight", (Hc,), "g"), (f"{p}.embeddings.LayerNorm.bias", (Hc,), "b")]
    for i in range(a["ce_layers"]):
        s += layer(f"{p}.encoder.layer.{i}", Hc, Ic, False)
    s += [("reranker.classifier1.weight", (1, Hc), "w"), ("reranker.classifier1.bias", (1,), "b"),
          ("reranker.classifier2.weight", (1, Hc), "w"), ("reranker.classifier2.bias", (1,), "b")]
    return s


VIT_PREFIX = "context_vision_encoder.vision_model.vision_model"   # FLMRVisionModel -> CLIPVisionModel -> transformer


def vit_weight_spec(a: dict) -> List[tuple]:
    """CLIP vision tower tensors under the reference's state_dict keys (modeling_flmr.py:1684-1757)."""
    Vh, Iv, ps, v = a["vision_hidden"], a["vit_intermediate"], a["vit_patch_size"], VIT_PREFIX
    s = [(f"{v}.embeddings.class_embedding", (Vh,), "e"),
         (f"{v}.embeddings.patch_embedding.weight", (Vh, 3, ps, ps), "w"),
         (f"{v}.embeddings.position_embedding.weight", (a["n_patches"] + 1, Vh), "e"),
         (f"{v}.pre_layrnorm.weight", (Vh,), "g"), (f"{v}.pre_layrnorm.bias", (Vh,), "b")]
    for i in range(a["vit_layers"]):
        l = f"{v}.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += [(f"{l}.self_attn.{n}.weight", (Vh, Vh), "w"), (f"{l}.self_attn.{n}.bias", (Vh,), "b")]
        s += [(f"{l}.layer_norm1.weight", (Vh,), "g"), (f"{l}.layer_norm1.bias", (Vh,), "b"),
              (f"{l}.mlp.fc1.weight", (Iv, Vh), "w"), (f"{l}.mlp.fc1.bias", (Iv,), "b"),
              (f"{l}.mlp.fc2.weight", (Vh, Iv), "w"), (f"{l}.mlp.fc2.bias", (Vh,), "b"),
              (f"{l}.layer_norm2.weight", (Vh,), "g"), (f"{l}.layer_norm2.bias", (Vh,), "b")]
    return s


def synthetic_state_dict(a: dict, seed: int = 0, hf_init: bool = True, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights (HF init: N(0, 0.02) matrices/embeddings, LN 1/0, zero biases —
    modeling_flmr.py:199-214) for benchmarks: there are no checkpoints in the build environment.
    Each tensor has its own generator seeded by (seed, index), identical to the test oracle's scheme so the
    CPU baseline can be fed the very same weights.  `gain` > 1 widens the Linear matrices only (the test suite's weight generator
    does the same): at 2.5 attention is peaked and a candidate list spreads over ~0.2 in logit, as in tests/golden c3_sep."""
    w = {}
    for idx, (name, shape, kind) in enumerate(weight_spec(a)):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if kind in ("w", "e"):
            t = torch.randn(shape, generator=g) * (0.02 * (gain if kind == "w" else 1.0))
        elif kind == "g":
            t = torch.ones(shape) if hf_init else 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape) if hf_init else 0.05 * torch.randn(shape, generator=g)
        w[name] = t
    return w


class RerankEngine:
    """Owns one `rr_handle` (one model replica on one GPU)."""

    def __init__(self, arch: dict, device: Optional[torch.device] = None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("rmr_amd needs an MI355X (gfx950) visible to HIP; there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.arch = dict(arch)
        c = L.RRConfig()
        c.abi_version = L.RR_ABI_VERSION
        for k in ("vocab_size", "hidden", "layers", "heads", "intermediate", "max_pos", "type_vocab", "li_dim",
                  "ce_hidden", "ce_layers", "ce_heads", "ce_intermediate", "ce_max_pos", "has_vision",
                  "vision_hidden", "prefix_len", "n_patches", "map_layers", "cross_attn_len"):
            setattr(c, k, int(arch[k]))
        c.ln_eps = float(arch["ln_eps"])
        if arch["loss_fn"] not in L.LOSS_KINDS:
            raise ValueError(f"Unknown loss function {arch['loss_fn']}")        # utils.py:222-223
        c.loss_kind = L.LOSS_KINDS[arch["loss_fn"]]
        c.pos_weight = float("nan") if arch.get("pos_weight") is None else float(arch["pos_weight"])
        cd = arch.get("compute_dtype", "bf16")
        if cd not in L.COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(L.COMPUTE_DTYPES)}, got {cd!r}")
        c.compute_dtype = L.COMPUTE_DTYPES[cd]
        mk = arch.get("model_kind", "full_context")
        if mk not in L.MODEL_KINDS:
            raise ValueError(f"model_kind must be one of {sorted(L.MODEL_KINDS)}, got {mk!r}")
        c.model_kind = L.MODEL_KINDS[mk]
        for k, v in VIT_DEFAULTS.items():
            setattr(c, k, int(arch.get(k, v)))
        c.fp8 = int(bool(arch.get("fp8", 0)))      # BASELINE configs[4]: e4m3 QKV / FFN-up GEMMs (rr_config.fp8)
        c.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        L.check(self.lib.rr_create(C.byref(c), C.byref(h)), None, "rr_create")
        self.h = h
        self.finalized = False

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                self.lib.rr_destroy(h)
            except Exception:
                pass
            self.h = None

    # ---- weights ------------------------------------------------------------------------------
    def required_weight_names(self) -> List[str]:
        n = self.lib.rr_num_required_weights(self.h)
        return [self.lib.rr_required_weight_name(self.h, i).decode() for i in range(n)]

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False, prefix: str = "") -> List[str]:
        """Feed a reference-named state_dict (Lightning ckpt keys carry a `reranker.` executor prefix:
        pass prefix="reranker.").  Unknown keys are ignored like load_state_dict(strict=False)
        (Reranker_base_executor.py:366-381); missing required tensors raise KeyError."""
        unexpected = []
        for k, t in sd.items():
            if prefix:
                if not k.startswith(prefix):
                    unexpected.append(k)
                    continue
                k = k[len(prefix):]
            t = t.detach().to("cpu").contiguous()
            if t.dtype == torch.float32:
                dt = L.RR_F32
            elif t.dtype == torch.bfloat16:
                dt = L.RR_BF16
            elif t.dtype == torch.float16:
                dt = L.RR_F16
            else:
                raise ValueError(f"{k}: unsupported dtype {t.dtype}")
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            known = C.c_int(0)
            L.check(self.lib.rr_load_weight(self.h, k.encode(), t.data_ptr(), dt, t.dim(), shape, C.byref(known)),
                    self.h, "rr_load_weight")
            if not known.value:
                unexpected.append(k)
        if strict and unexpected:
            raise KeyError(f"unexpected keys: {unexpected[:5]}...")
        L.check(self.lib.rr_finalize_weights(self.h), self.h, "rr_finalize_weights")
        self.finalized = True
        return unexpected

    # ---- forward ------------------------------------------------------------------------------
    def forward_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                    token_type_ids: Optional[torch.Tensor], Bq: int, K: int,
                    image_cls: Optional[torch.Tensor] = None, image_patches: Optional[torch.Tensor] = None,
                    labels: Optional[torch.Tensor] = None, want_scores: bool = False, want_order: bool = False,
                    pair_range: Optional[Sequence[int]] = None, want_loss: bool = True):
        """One pass over the tokenised pair batch.  All tensors live on `self.device`.
        Returns dict(logits [N] fp32, logits2 [N], loss 0-dim | None, scores | None, order [Bq,K] | None)."""
        dev = self.device
        N = input_ids.shape[0]
        assert N == Bq * K, f"expanded batch {Bq}*{K} != {N}"                 # rerank_model.py:527
        S = input_ids.shape[1]
        for t in (input_ids, attention_mask) + ((token_type_ids,) if token_type_ids is not None else ()):
            if t.dtype != torch.int64 or t.device != dev or tuple(t.shape) != (N, S):
                raise ValueError("input_ids/attention_mask/token_type_ids must be int64 [N,S] on the model device")
        if labels is not None:
            assert labels.numel() == N, "len(labels) != expanded batch size"   # rerank_model.py:528-529
            labels = labels.to(device=dev, dtype=torch.float32).contiguous()
        if image_cls is not None:
            image_cls = image_cls.to(device=dev, dtype=torch.float32).contiguous()
            image_patches = image_patches.to(device=dev, dtype=torch.float32).contiguous()
            if image_cls.shape[0] != Bq or image_patches.shape[0] != Bq:
                raise AssertionError("image features must be per query: [Bq, ...]")
        pb, pe = (0, N) if pair_range is None else (int(pair_range[0]), int(pair_range[1]))
        full = pb == 0 and pe == N
        logits = torch.empty(N, dtype=torch.float32, device=dev)
        logits2 = torch.empty(N, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev) if (full and want_loss) else None
        scores = torch.empty(N, dtype=torch.float32, device=dev) if (full and want_scores) else None
        order = torch.empty((Bq, K), dtype=torch.int32, device=dev) if (full and want_order) else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(self.lib.rr_forward(self.h, L.ptr(input_ids), L.ptr(attention_mask), L.ptr(token_type_ids),
                                    L.ptr(image_cls), L.ptr(image_patches), Bq, K, S, L.ptr(labels), pb, pe,
                                    L.ptr(logits), L.ptr(logits2), L.ptr(loss), L.ptr(scores), L.ptr(order),
                                    stream), self.h, "rr_forward")
        return dict(logits=logits, logits2=logits2, loss=loss, scores=scores, order=order)

    def forward_ids_bucketed(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids: Optional[torch.Tensor],
                             Bq: int, K: int, image_cls: Optional[torch.Tensor] = None,
                             image_patches: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                             buckets: Sequence[int] = (128, 256, 384), want_scores: bool = False,
                             want_order: bool = False):
        """The same result as `forward_ids` on right-padded pairs, computed per LENGTH BUCKET: the reference pads every pair
        to max_decoder_source_length (utils.py:157-165) and real passages are far shorter, so the pairs are grouped by the
        smallest bucket length that holds their last non-pad position, each group runs with that row length (its GEMMs,
        LayerNorms and query rows shrink in proportion; rr_set_padded_seq_len keeps the cross-encoder's vision positions) and
        the head / top-K run once on the reassembled logits (rr_head).  One device -> host copy of N lengths per call.
        Bucket sizes: every group is its own forward (~135 launches) over fewer pairs, so few, wide buckets win: measured
        on 800 pairs of length U[64, 512] (bench.py --regime realistic --bucketed): (128, 256, 384) 71.5 ms, (192, 320) 74.8,
        (256,) 77.8, five buckets 77.5, seven 82.1, against 94.5 ms padded.
        Text-only: logits bit-identical to forward_ids WHILE every bucket and the padded call lie on the same side of two size
        thresholds (include/rerank_mi355.h, rr_set_padded_seq_len: the 1 024-workgroup attention schedule switch and the
        128-tile switch between the fp32 and the (hi, lo) residual stream, about 11k rows at hidden 768); across them, and with
        vision tokens (other key-tile cuts in the cross-encoder's attention), equal to the parity tolerance — 1e-3 in fp16,
        measured ~2e-4 — not to the bit.  Returns the dict of forward_ids."""
        dev = self.device
        N, S = input_ids.shape
        assert N == Bq * K
        cols = torch.arange(1, S + 1, device=dev)
        used = (input_ids != 0) | (attention_mask != 0)
        lens = (used * cols).amax(1)                                   # 1 + index of the last non-pad position
        # with vision tokens the mapping network attends to the first cross_attn_len text rows: no bucket below that (the
        # library refuses it: RR_ERR_BAD_SHAPE)
        floor = min(S, int(self.arch.get("cross_attn_len", 32))) if image_cls is not None else 1
        sizes = sorted({max(int(b), floor) for b in buckets if 0 < int(b) < S and max(int(b), floor) < S}) + [S]
        which = torch.bucketize(lens, torch.tensor(sizes, device=dev))  # smallest bucket with size >= len
        counts = torch.bincount(which, minlength=len(sizes)).cpu().tolist()
        logits = torch.empty(N, dtype=torch.float32, device=dev)
        logits2 = torch.empty(N, dtype=torch.float32, device=dev)
        two = self.arch["loss_fn"] == "2H_BCE"
        L.check(self.lib.rr_set_padded_seq_len(self.h, S), self.h, "rr_set_padded_seq_len")
        try:
            order_by_bucket = torch.argsort(which, stable=True)
            o = 0
            for b, n in enumerate(counts):
                if n == 0:
                    continue
                idx = order_by_bucket[o: o + n]
                o += n
                Sb = sizes[b]
                sub = [t.index_select(0, idx)[:, :Sb].contiguous() if t is not None else None
                       for t in (input_ids, attention_mask, token_type_ids)]
                cls_b = pat_b = None
                if image_cls is not None:                               # per pair here: the group mixes candidates of several queries
                    q = torch.div(idx, K, rounding_mode="floor")
                    cls_b, pat_b = image_cls.index_select(0, q), image_patches.index_select(0, q)
                r = self.forward_ids(sub[0], sub[1], sub[2], n, 1, cls_b, pat_b, None, want_loss=False)
                logits.index_copy_(0, idx, r["logits"])
                if two:
                    logits2.index_copy_(0, idx, r["logits2"])
        finally:
            self.lib.rr_set_padded_seq_len(self.h, 0)
        out = self.head(logits, logits2 if two else None, labels, Bq, K, want_scores=want_scores, want_order=want_order)
        out["logits"], out["logits2"] = logits, logits2
        out["bucket_rows"] = sum(n * sizes[b] for b, n in enumerate(counts))      # rows actually computed (N * S when nothing fits a bucket)
        return out

    def forward_ids_packed(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids: Optional[torch.Tensor],
                           Bq: int, K: int, image_cls: Optional[torch.Tensor] = None,
                           image_patches: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                           granule: int = 16, want_scores: bool = False, want_order: bool = False,
                           lengths: Optional[Sequence[int]] = None, segment_cost_rows: int = 0):
        """The same result as `forward_ids` on right-padded pairs, computed over PACKED rows (rr_forward_packed): the pairs are
        grouped by their length rounded up to a multiple of `granule` and laid out group after group, so that every GEMM /
        LayerNorm pass of a layer runs once over the rows that exist — the reference pads every pair to
        max_decoder_source_length (utils.py:157-165) — while attention runs once per group.  Against forward_ids_bucketed
        (one whole forward per group): the same rows at granule 128, but one large GEMM launch and ONE attention launch per
        layer for all groups instead of one per group, which is what lets the granule shrink (measured on lengths U[64, 512]:
        granule 64 / 32 / 16 / 8 -> 62.1 / 60.6 / 59.7 / 62.5 ms against 93.2 padded).  Logits: bit-identical to forward_ids_bucketed on the same groups, i.e.
        to forward_ids for text-only models.  `lengths`: the pairs' token counts (1 + index of the last non-pad position) as
        the HOST knows them from the tokenizer (pair_inputs.prepare_full_context_inputs keeps them); without it they are
        derived on the device and the group counts cost one device -> host copy per call, which drains the stream between
        two forwards.  `segment_cost_rows`: merge neighbouring lengths where a segment's fixed launches cost more than the rows
        the merge pads (pair_inputs.group_pairs_by_length).  Returns the dict of forward_ids plus `packed_rows`, `packed_segments`."""
        dev = self.device
        N, S = input_ids.shape
        assert N == Bq * K and granule > 0
        floor = int(self.arch.get("cross_attn_len", 32)) if image_cls is not None else 1   # the mapping network's cross-attention window
        if lengths is None:                                            # derived on the device: one device -> host copy
            cols = torch.arange(1, S + 1, device=dev)
            lengths = (((input_ids != 0) | (attention_mask != 0)) * cols).amax(1).cpu().numpy()
        order_h, seg_n, seg_len = group_pairs_by_length(lengths, S, granule, floor, segment_cost_rows)
        assert len(order_h) == N, "one length per pair"
        order = torch.from_numpy(order_h).to(dev, non_blocking=True)
        parts = [[], [], []]
        o = 0
        for n, sb in zip(seg_n, seg_len):
            idx = order[o: o + n]
            o += n
            for dst, t in zip(parts, (input_ids, attention_mask, token_type_ids)):
                if t is not None:
                    dst.append(t.index_select(0, idx)[:, :sb].reshape(-1))
        ids_p, am_p = torch.cat(parts[0]), torch.cat(parts[1])
        tt_p = torch.cat(parts[2]) if token_type_ids is not None else None
        cls_p = pat_p = None
        if image_cls is not None:                                       # per pair: a group mixes candidates of several queries
            q = torch.div(order, K, rounding_mode="floor")
            cls_p = image_cls.index_select(0, q).float().contiguous()
            pat_p = image_patches.index_select(0, q).float().contiguous()
        two = self.arch["loss_fn"] == "2H_BCE"
        lp = torch.empty(N, dtype=torch.float32, device=dev)
        lp2 = torch.empty(N, dtype=torch.float32, device=dev) if two else None
        sn = (C.c_int32 * len(seg_n))(*seg_n)
        sl = (C.c_int32 * len(seg_n))(*seg_len)
        L.check(self.lib.rr_forward_packed(self.h, L.ptr(ids_p), L.ptr(am_p), L.ptr(tt_p), L.ptr(cls_p), L.ptr(pat_p), len(seg_n),
                                           sn, sl, S, L.ptr(lp), L.ptr(lp2), torch.cuda.current_stream(dev).cuda_stream),
                self.h, "rr_forward_packed")
        logits = torch.empty_like(lp).index_copy_(0, order, lp)
        logits2 = torch.empty_like(lp).index_copy_(0, order, lp2) if two else torch.empty(N, dtype=torch.float32, device=dev)
        out = self.head(logits, logits2 if two else None, labels, Bq, K, want_scores=want_scores, want_order=want_order)
        out["logits"], out["logits2"] = logits, logits2
        out["packed_rows"] = sum(n * s for n, s in zip(seg_n, seg_len))
        out["packed_segments"] = len(seg_n)
        return out

    def activation_range_exceeded(self, reset: bool = True) -> bool:
        """True when, since the last reset, a pre-LayerNorm residual row came within a factor 2 of the fp16 range (or was
        not finite) — rr_activation_range_flag; synchronises the current stream, so call it once per batch group, not per
        forward.  The remedy is an engine with compute_dtype="bf16" (DESIGN.md "Numerics")."""
        flag = C.c_int(0)
        L.check(self.lib.rr_activation_range_flag(self.h, int(reset), C.byref(flag), torch.cuda.current_stream(self.device).cuda_stream),
                self.h, "rr_activation_range_flag")
        return flag.value != 0

    def encode_image(self, pixel_values: torch.Tensor):
        """CLIP vision tower (rr_encode_image): pixel_values [B,3,IS,IS] -> (last_hidden_state[:,0] [B,Vh],
        hidden_states[-2][:,1:] [B,np,Vh]) — what rerank_model.py:408-411,424-426 takes from context_vision_encoder."""
        a, dev = self.arch, self.device
        IS = int(a.get("vit_image_size", 224))
        if pixel_values.dim() == 5:                                   # [B,1,3,H,W] as the datasets deliver it
            pixel_values = pixel_values.reshape(-1, *pixel_values.shape[2:])
        if tuple(pixel_values.shape[1:]) != (3, IS, IS):
            raise AssertionError(f"pixel_values must be [B,3,{IS},{IS}], got {tuple(pixel_values.shape)}")
        px = pixel_values.to(device=dev, dtype=torch.float32).contiguous()
        B = px.shape[0]
        cls = torch.empty((B, a["vision_hidden"]), dtype=torch.float32, device=dev)
        patches = torch.empty((B, a["n_patches"], a["vision_hidden"]), dtype=torch.float32, device=dev)
        L.check(self.lib.rr_encode_image(self.h, L.ptr(px), B, L.ptr(cls), L.ptr(patches),
                                         torch.cuda.current_stream(dev).cuda_stream), self.h, "rr_encode_image")
        return cls, patches

    def forward_joint(self, joint_input_ids: torch.Tensor, joint_attention_mask: torch.Tensor, Bq: int, K: int,
                      query_len: int, image_cls: torch.Tensor, image_patches: torch.Tensor,
                      instruction_token_id: Optional[int] = None, want_scores: bool = False, want_order: bool = False,
                      pair_range: Optional[Sequence[int]] = None, want_loss: bool = True,
                      preflmr_scores: Optional[torch.Tensor] = None, fusion_multiplier: float = 1.0):
        """RerankModel.forward semantics on the assembled joint sequence (see rr_forward_joint); `preflmr_scores`
        [N, S, query_len + image tokens] switches the PreFLMR attention fusion on (rr_forward_joint_fusion)."""
        dev = self.device
        N, S = joint_input_ids.shape
        assert N == Bq * K
        f32 = dict(device=dev, dtype=torch.float32)
        cls = patches = None
        if image_cls is not None:
            cls, patches = image_cls.to(**f32).contiguous(), image_patches.to(**f32).contiguous()
        pb, pe = (0, N) if pair_range is None else (int(pair_range[0]), int(pair_range[1]))
        full = pb == 0 and pe == N
        logits, logits2 = torch.empty(N, **f32), torch.empty(N, **f32)
        loss = torch.empty((), **f32) if (full and want_loss) else None
        scores = torch.empty(N, **f32) if (full and want_scores) else None
        order = torch.empty((Bq, K), dtype=torch.int32, device=dev) if (full and want_order) else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        instr = -1 if instruction_token_id is None else int(instruction_token_id)
        if preflmr_scores is not None:
            P = self.arch["prefix_len"] + self.arch["n_patches"]
            ps = preflmr_scores.to(**f32).contiguous()
            if tuple(ps.shape) != (N, S, int(query_len) + P):                                   # rerank_model.py:280-284
                raise AssertionError(f"preflmr_scores must be [{N}, {S}, {int(query_len) + P}], got {tuple(ps.shape)}")
            L.check(self.lib.rr_forward_joint_fusion(self.h, L.ptr(joint_input_ids.contiguous()),
                                                     L.ptr(joint_attention_mask.contiguous()), L.ptr(cls), L.ptr(patches),
                                                     L.ptr(ps), float(fusion_multiplier), Bq, K, S, int(query_len), instr, pb,
                                                     pe, L.ptr(logits), L.ptr(logits2), L.ptr(loss), L.ptr(scores),
                                                     L.ptr(order), stream), self.h, "rr_forward_joint_fusion")
        else:
            L.check(self.lib.rr_forward_joint(self.h, L.ptr(joint_input_ids.contiguous()),
                                              L.ptr(joint_attention_mask.contiguous()), L.ptr(cls), L.ptr(patches), Bq, K, S,
                                              int(query_len), instr, pb, pe, L.ptr(logits), L.ptr(logits2), L.ptr(loss),
                                              L.ptr(scores), L.ptr(order), stream), self.h, "rr_forward_joint")
        return dict(logits=logits, logits2=logits2, loss=loss, scores=scores, order=order)

    def forward_interaction(self, query_li: torch.Tensor, context_li: torch.Tensor, query_mask: torch.Tensor,
                            context_mask: torch.Tensor, Bq: int, K: int, labels: Optional[torch.Tensor] = None,
                            want_scores: bool = False, want_order: bool = False,
                            pair_range: Optional[Sequence[int]] = None, want_loss: bool = True,
                            preflmr_scores: Optional[torch.Tensor] = None, fusion_multiplier: float = 1.0):
        """Interaction rerankers: late-interaction tensors [Bq,Lq,D] / [N,Lc,D] and 0/1 masks [Bq,Lq] / [N,Lc];
        `preflmr_scores` [N, Lc, Lq] switches the attention fusion on (rr_forward_interaction_fusion)."""
        dev = self.device
        N = context_li.shape[0]
        assert N == Bq * K and query_li.shape[0] == Bq, \
            f"{tuple(query_li.shape)}, {tuple(context_li.shape)}, {K - 1}"        # interaction_rerank_model.py:123
        Lq, Lc = query_li.shape[1], context_li.shape[1]
        f32 = dict(device=dev, dtype=torch.float32)
        query_li, context_li = query_li.to(**f32).contiguous(), context_li.to(**f32).contiguous()
        query_mask = query_mask.reshape(Bq, Lq).to(**f32).contiguous()
        context_mask = context_mask.reshape(N, Lc).to(**f32).contiguous()
        if labels is not None:
            assert labels.numel() == N
            labels = labels.to(**f32).contiguous()
        pb, pe = (0, N) if pair_range is None else (int(pair_range[0]), int(pair_range[1]))
        full = pb == 0 and pe == N
        logits = torch.empty(N, **f32)
        logits2 = torch.empty(N, **f32)
        loss = torch.empty((), **f32) if (full and want_loss) else None
        scores = torch.empty(N, **f32) if (full and want_scores) else None
        order = torch.empty((Bq, K), dtype=torch.int32, device=dev) if (full and want_order) else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        if preflmr_scores is not None:
            ps = preflmr_scores.to(**f32).contiguous()
            if tuple(ps.shape) != (N, Lc, Lq):
                raise AssertionError(f"preflmr_scores must be [{N}, {Lc}, {Lq}], got {tuple(ps.shape)}")
            L.check(self.lib.rr_forward_interaction_fusion(self.h, L.ptr(query_li), L.ptr(context_li), L.ptr(query_mask),
                                                           L.ptr(context_mask), L.ptr(ps), float(fusion_multiplier), Bq, K,
                                                           Lq, Lc, L.ptr(labels), pb, pe, L.ptr(logits), L.ptr(logits2),
                                                           L.ptr(loss), L.ptr(scores), L.ptr(order), stream),
                    self.h, "rr_forward_interaction_fusion")
        else:
            L.check(self.lib.rr_forward_interaction(self.h, L.ptr(query_li), L.ptr(context_li), L.ptr(query_mask),
                                                    L.ptr(context_mask), Bq, K, Lq, Lc, L.ptr(labels), pb, pe,
                                                    L.ptr(logits), L.ptr(logits2), L.ptr(loss), L.ptr(scores),
                                                    L.ptr(order), stream), self.h, "rr_forward_interaction")
        return dict(logits=logits, logits2=logits2, loss=loss, scores=scores, order=order)

    def head(self, logits: torch.Tensor, logits2: Optional[torch.Tensor], labels: Optional[torch.Tensor], Bq: int,
             K: int, want_scores: bool = False, want_order: bool = True):
        """Scoring head on complete logits (after the cross-rank all-gather)."""
        dev = self.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        scores = torch.empty(Bq * K, dtype=torch.float32, device=dev) if want_scores else None
        order = torch.empty((Bq, K), dtype=torch.int32, device=dev) if want_order else None
        if labels is not None:
            labels = labels.to(device=dev, dtype=torch.float32).contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(self.lib.rr_head(self.h, L.ptr(logits), L.ptr(logits2), L.ptr(labels), Bq, K, L.ptr(loss),
                                 L.ptr(scores), L.ptr(order), stream), self.h, "rr_head")
        return dict(loss=loss, scores=scores, order=order)

    # ---- debugging / profiling ------------------------------------------------------------------
    def set_debug(self, on: bool):
        L.check(self.lib.rr_set_debug(self.h, int(on)), self.h)

    def debug_read(self, name: str, numel: int) -> torch.Tensor:
        out = torch.empty(numel, dtype=torch.float32)
        n = self.lib.rr_debug_read(self.h, name.encode(), out.data_ptr(), numel)
        if n < 0:
            L.check(int(n), self.h, "rr_debug_read")
        return out[:n]

    def set_profiling(self, on: bool):
        L.check(self.lib.rr_set_profiling(self.h, int(on)), self.h)

    def get_profile(self, reset: bool = True) -> Dict[str, dict]:
        p = L.RRProfile()
        L.check(self.lib.rr_get_profile(self.h, C.byref(p), int(reset)), self.h, "rr_get_profile")
        return {k: dict(ms=p.ms[i], launches=p.launches[i], flops=p.flops[i], bytes=p.bytes[i])
                for i, k in enumerate(L.KERNEL_CLASSES)}

    def workspace_bytes(self, n_pairs: int, S: int) -> int:
        return int(self.lib.rr_workspace_bytes(self.h, n_pairs, S))

    def set_option(self, key: str, value: int):
        """Pin a numerics option of THIS engine (rr_set_option: "ln_lite", "ln_fold", "resid_split", "resid_lo8", "ce_cls_only",
        "fp8_ffn_down", "fp8_first_layer", "fp8_qkv", "attn_fixed_ref"); -1 = follow the process-wide diagnostic switch again."""
        L.check(self.lib.rr_set_option(self.h, key.encode(), int(value)), self.h, "rr_set_option")

    def get_option(self, key: str) -> int:
        import ctypes
        v = ctypes.c_int(0)
        L.check(self.lib.rr_get_option(self.h, key.encode(), ctypes.byref(v)), self.h, "rr_get_option")
        return int(v.value)

    def reserve(self, n_pairs: int, n_queries: int, len_a: int, len_b: int = 0, with_fusion: bool = False,
                packed: bool = False):
        """Allocate everything a forward of at most this shape needs (rr_reserve) on the current stream: afterwards the
        forward neither allocates nor synchronises (a precondition for capturing it into a hipGraph).  `packed`: for
        forward_ids_packed / forward_ids_bucketed, whose image features are per PAIR (n_queries = n_pairs, as
        include/rerank_mi355.h documents for rr_forward_packed)."""
        if packed:
            n_queries = n_pairs
        L.check(self.lib.rr_reserve(self.h, int(n_pairs), int(n_queries), int(len_a), int(len_b), int(with_fusion),
                                    torch.cuda.current_stream(self.device).cuda_stream), self.h, "rr_reserve")


class _FrozenStub(torch.nn.Module):
    """Placeholder for `context_vision_encoder`: the executor only iterates its `named_parameters()` to
    freeze them (Reranker_base_executor.py:204-207).  The CLIP ViT is upstream of this path."""


class FullContextRerankModel(torch.nn.Module):
    """Drop-in for the reference's `FullContextRerankModel` (rerank_model.py:515-591), inference only.

    `config` is the reference `reranker_config` (EasyDict/dict).  Extra, optional keys:
      `arch`            – dict overriding the FLMR/BERT architecture defaults (tests use tiny shapes)
      `tokenizer`       – any HF-style tokenizer (encode/decode/batch_encode_plus); required only for the
                          text call signature (no vocab file exists in the build environment)
      `vision_encoder`  – True: run the CLIP ViT-B/32 tower inside the library (rr_encode_image; the state_dict must
                          then hold `context_vision_encoder.vision_model.vision_model.*`)
      `image_feature_fn`– otherwise a callable pixel_values[Bq,3,224,224] -> (cls [Bq,Vh], patches [Bq,np,Vh]) that
                          wraps the reference-side `context_vision_encoder`
      `text_only`       – build without the vision weights (`text_only` module of the reference configs)
      `native_tokenizer`– True: assemble the pair inputs with the library's multi-threaded C++ WordPiece tokenizer
                          (rr_tok_prepare_pairs) built from `tokenizer`'s vocabulary instead of calling `tokenizer`
    """

    def __init__(self, config, state_dict: Optional[Dict[str, torch.Tensor]] = None, device=None):
        super().__init__()
        self.config = config
        arch = make_arch(config)
        if _get(config, "text_only", False):
            arch["has_vision"] = 0
        self.engine = RerankEngine(arch, device)
        self.max_query_length = _get(config, "max_query_length", 32)
        self.max_decoder_source_length = _get(config, "max_decoder_source_length", 512)
        self.max_context_length = self.max_decoder_source_length - self.max_query_length - 4   # HEAD_TOKEN_LEEWAY
        self.query_tokenizer = _get(config, "tokenizer", None)
        self.native_tokenizer = None
        if _get(config, "native_tokenizer", False):
            if self.query_tokenizer is None:
                raise ValueError("native_tokenizer needs config.tokenizer (for its vocabulary)")
            from .pair_inputs import NativePairTokenizer
            self.native_tokenizer = NativePairTokenizer(self.query_tokenizer,
                                                        do_lower_case=getattr(self.query_tokenizer, "do_lower_case", True))
        self.image_feature_fn = _get(config, "image_feature_fn", None)
        self.context_vision_encoder = _FrozenStub()
        if state_dict is not None:
            self.engine.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict: bool = False, prefix: str = ""):  # type: ignore[override]
        return self.engine.load_state_dict(state_dict, strict=strict, prefix=prefix)

    # tensor fast path (synthetic benchmarks, pre-tokenised callers)
    def forward_ids(self, input_ids, attention_mask, token_type_ids, num_negative_examples: int,
                    image_cls=None, image_patches=None, labels: Optional[List[float]] = None, **kw) -> RerankOutput:
        K = num_negative_examples + 1
        N = input_ids.shape[0]
        assert N % K == 0, "expanded batch size must be batch_size * (num_negative_examples + 1)"
        Bq = N // K
        arch = self.engine.arch
        if labels is not None:
            assert isinstance(labels, list), "Labels must be a list"                       # utils.py:232
            if arch["loss_fn"] == "negative_sampling":
                raise AssertionError("Labels should not be provided for negative sampling loss function")
            assert len(labels) == N                                                        # rerank_model.py:528-529
            labels_t = torch.tensor(labels, dtype=torch.float32, device=self.engine.device)
        else:
            labels_t = None
        r = self.engine.forward_ids(input_ids, attention_mask, token_type_ids, Bq, K, image_cls, image_patches,
                                    labels_t, **kw)
        logits = r["logits"]
        logits = logits.view(Bq, K) if arch["loss_fn"] == "negative_sampling" else logits.view(N, 1)
        out = RerankOutput(loss=r["loss"], logits=logits)
        for k in ("scores", "order", "logits2"):
            if r.get(k) is not None:
                out[k] = r[k]
        return out

    def forward(self, query_text_sequences, query_pixel_values, context_text_sequences, num_negative_examples,
                labels=None) -> RerankOutput:
        text_only = query_pixel_values is None
        batch_size = len(query_text_sequences)
        expanded = batch_size * (num_negative_examples + 1)
        assert expanded == len(context_text_sequences)                                     # rerank_model.py:527
        if labels:
            assert len(labels) == expanded
        if self.query_tokenizer is None:
            raise RuntimeError("text call signature needs config.tokenizer (an HF-style BERT tokenizer)")
        if self.native_tokenizer is not None:
            enc = self.native_tokenizer.prepare_full_context_inputs(
                list(query_text_sequences), list(context_text_sequences), self.max_query_length, self.max_context_length,
                self.max_decoder_source_length, num_negative_examples + 1, pin_memory=True)
        else:
            from .pair_inputs import prepare_full_context_inputs
            enc = prepare_full_context_inputs(query_text_sequences, context_text_sequences, self.query_tokenizer,
                                              self.max_query_length, self.max_context_length,
                                              self.max_decoder_source_length, num_negative_examples + 1)
        dev = self.engine.device
        cls = patches = None
        if not text_only:
            if self.image_feature_fn is not None:
                cls, patches = self.image_feature_fn(query_pixel_values)
            elif self.engine.arch.get("vit_layers", 0) > 0:
                cls, patches = self.engine.encode_image(query_pixel_values)
            else:
                raise NotImplementedError("query_pixel_values given but neither config.vision_encoder nor "
                                          "config.image_feature_fn (CLIP ViT) is set")
        return self.forward_ids(enc["input_ids"].to(dev), enc["attention_mask"].to(dev),
                                enc["token_type_ids"].to(dev), num_negative_examples, cls, patches,
                                labels if labels else None)


class InteractionRerankModel(torch.nn.Module):
    """Drop-in for the reference's `InteractionRerankModel` (interaction_rerank_model.py:86-166), inference only:
    `config.interaction_type` "MORES" selects the MORES stack (mores_model.py), anything else the CrossEncoder."""

    def __init__(self, config, state_dict: Optional[Dict[str, torch.Tensor]] = None, device=None):
        super().__init__()
        self.config = config
        kind = "mores" if _get(config, "interaction_type", "NORMAL") == "MORES" else "interaction"
        arch = make_arch(config, model_kind=kind, has_vision=0)
        self.engine = RerankEngine(arch, device)
        if state_dict is not None:
            self.engine.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict: bool = False, prefix: str = ""):  # type: ignore[override]
        return self.engine.load_state_dict(state_dict, strict=strict, prefix=prefix)

    def forward(self, query_late_interaction, context_late_interaction, num_negative_examples, query_mask,
                context_mask, preflmr_scores=None, fusion_multiplier=1, labels=None, **kw) -> RerankOutput:
        K = num_negative_examples + 1
        Bq = query_late_interaction.size(0)
        N = context_late_interaction.size(0)
        assert Bq * K == N, f"{query_late_interaction.shape}, {context_late_interaction.shape}, {num_negative_examples}"
        arch = self.engine.arch
        labels_t = None
        if labels is not None:
            assert isinstance(labels, list), "Labels must be a list"
            if arch["loss_fn"] == "negative_sampling":
                raise AssertionError("Labels should not be provided for negative sampling loss function")
            labels_t = torch.tensor(labels, dtype=torch.float32, device=self.engine.device)
        r = self.engine.forward_interaction(query_late_interaction, context_late_interaction, query_mask, context_mask,
                                            Bq, K, labels_t, preflmr_scores=preflmr_scores,
                                            fusion_multiplier=float(fusion_multiplier), **kw)
        logits = r["logits"]
        logits = logits.view(Bq, K) if arch["loss_fn"] == "negative_sampling" else logits.view(N, 1)
        out = RerankOutput(loss=r["loss"], logits=logits)
        for k in ("scores", "order", "logits2"):
            if r.get(k) is not None:
                out[k] = r[k]
        return out


class RerankModel(torch.nn.Module):
    """Drop-in for the reference's `RerankModel` (rerank_model.py:76-331; the "softmax"/2-head variant), inference
    only.  Extra optional config keys: `arch`, `image_feature_fn` (pixel_values -> (cls, patches)),
    `instruction_token_id` (id of `mask_instruction_token`, rerank_model.py:161-169; None = no instruction masking)."""

    def __init__(self, config, state_dict: Optional[Dict[str, torch.Tensor]] = None, device=None):
        super().__init__()
        self.config = config
        self.engine = RerankEngine(make_arch(config), device)
        self.image_feature_fn = _get(config, "image_feature_fn", None)
        self.instruction_token_id = _get(config, "instruction_token_id", None)
        self.context_vision_encoder = _FrozenStub()
        if state_dict is not None:
            self.engine.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict: bool = False, prefix: str = ""):  # type: ignore[override]
        return self.engine.load_state_dict(state_dict, strict=strict, prefix=prefix)

    def forward(self, query_input_ids, query_attention_mask, query_pixel_values, context_input_ids,
                context_attention_mask, num_negative_examples, preflmr_scores=None, fusion_multiplier=1, labels=None,
                image_features=None, **kw) -> RerankOutput:
        if query_pixel_values is None and image_features is None:
            raise NotImplementedError("text_only is not implemented for this model")        # rerank_model.py:184-185
        K = num_negative_examples + 1
        Bq = query_input_ids.size(0)
        N = Bq * K
        assert N == context_input_ids.size(0)                                               # :188
        if labels:
            assert len(labels) == N                                                         # :189-190
        ql, S = query_input_ids.size(1), context_input_ids.size(1)
        assert S == self.engine.arch["max_pos"]                                             # :202
        dev = self.engine.device
        # joint sequence exactly as :191-224 builds it (index plumbing on device tensors, no arithmetic)
        q_ids = query_input_ids.to(dev).repeat_interleave(K, dim=0)
        q_am = query_attention_mask.to(dev).repeat_interleave(K, dim=0)
        joint_ids = torch.cat([q_ids, context_input_ids.to(dev)[:, 2:2 - ql]], dim=1).to(torch.int64).contiguous()
        joint_am = torch.cat([q_am, context_attention_mask.to(dev)[:, 2:2 - ql]], dim=1).to(torch.int64).contiguous()
        if image_features is not None:
            cls, patches = image_features
        else:
            if self.image_feature_fn is not None:
                cls, patches = self.image_feature_fn(query_pixel_values)
            elif self.engine.arch.get("vit_layers", 0) > 0:
                cls, patches = self.engine.encode_image(query_pixel_values)
            else:
                raise NotImplementedError("query_pixel_values given but neither config.vision_encoder nor "
                                          "config.image_feature_fn (CLIP ViT) is set")
        r = self.engine.forward_joint(joint_ids, joint_am, Bq, K, ql, cls, patches, self.instruction_token_id,
                                      preflmr_scores=preflmr_scores, fusion_multiplier=float(fusion_multiplier), **kw)
        out = RerankOutput(loss=r["loss"], logits=r["logits"].view(N, 1))
        for k in ("scores", "order", "logits2"):
            if r.get(k) is not None:
                out[k] = r[k]
        return out
