"""CPU oracle for the cross-encoder rerank forward.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, not the product: only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  The product path
(`rmr_amd`, `librerank_mi355.so`) never routes through it and has no CPU fallback.

It restates, in plain fp32 `torch` tensor arithmetic on the CPU (no HuggingFace
module, no autocast, no fused kernels), the algorithm of the reference's rerank
hot path.  Every function cites the reference file:line (relative to
`/root/reference/`) it follows.  The transformer-block arithmetic lives in a
third-party dependency that is not vendored in the reference
(`transformers==4.38.2`, `README.md:90-91`; classes `BertModel`, `BertEncoder`,
`BertLayer`): its published algorithm (post-LN BERT, erf-GELU, eps 1e-12,
softmax(QK^T/sqrt(dh) + additive mask) V) is restated here and anchored on the
reference's own call sites.

Parity pin: the reference has NO golden vectors / known-answer tests for this path
("parity unpinned by the reference").  The restatement is pinned
instead against the stock HuggingFace `BertModel`/`BertEncoder` (eager attention)
assembled exactly as the reference assembles them; `tests/golden/make_golden.py`
is the generator (runs only in the build container) and `tests/golden/*.npz` are
the committed vectors.

Weights are a flat dict {reference state_dict key (without the `reranker.` executor
prefix) -> fp32 tensor}, e.g.
  context_text_encoder.bert_model.encoder.layer.3.attention.self.query.weight
  reranker.bert_model.embeddings.position_embeddings.weight
  cross_encoder_input_mapping.weight
(see SURVEY.md §5 "checkpoint / resume" for the naming).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
FMIN = torch.finfo(torch.float32).min


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Architecture of the path.  Defaults = monoPreFLMR-B / bert-base-uncased
    (`configuration_flmr.py:220-236,332`, `monoBERT_pointwise.jsonnet:111-122`)."""
    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    li_dim: int = 128                      # late-interaction dim (configuration_flmr.py:332)
    # cross encoder (rerank_model.py:89-101; bert-base-uncased shape, Lc layers, longer pos table)
    ce_hidden: int = 768
    ce_layers: int = 1
    ce_heads: int = 12
    ce_intermediate: int = 3072
    ce_max_pos: int = 750
    # vision side (modeling_flmr.py:603-664)
    vision_hidden: int = 768
    prefix_len: int = 32                   # mapping_network_prefix_length
    n_patches: int = 49                    # ViT-B/32: 7x7 patches
    map_layers: int = 1                    # transformer_mapping_num_hidden_layers
    cross_attn_len: int = 32               # transformer_mapping_cross_attention_length
    loss_fn: str = "BCE"                   # BCE | 2H_BCE | negative_sampling (utils.py:208-224)
    pos_weight: Optional[float] = None
    # CLIP vision tower (FLMRVisionConfig defaults = openai/clip-vit-base-patch32, configuration_flmr.py:90-104)
    vit_layers: int = 12
    vit_heads: int = 12
    vit_intermediate: int = 3072
    vit_image_size: int = 224
    vit_patch_size: int = 32


# --------------------------------------------------------------------------- blocks
def linear(x: Tensor, w: Dict[str, Tensor], name: str, mm=None) -> Tensor:
    """y = x W^T + b (torch.nn.Linear).  `mm` lets tests swap in a bf16-rounding
    matmul to emulate the device's rounding points; default = fp32."""
    W = w[name + ".weight"]
    b = w.get(name + ".bias")
    if mm is not None and name.endswith(_QUERY_NAMES):
        # device: the packed query weight is 16bit(W * log2(e)/sqrt(dh)), its bias b * the same factor (rr_api.hip
        # pack_layer); dividing back keeps this function's contract (unscaled q) with the device's rounding of W
        y = mm(x, W * _QSCALE) * (1.0 / _QSCALE)
    else:
        y = (mm(x, W) if mm is not None else x @ W.t())
    return y if b is None else y + b


def layer_norm(x: Tensor, w: Dict[str, Tensor], name: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], eps)


def gelu_erf(x: Tensor) -> Tensor:
    """HF `hidden_act="gelu"` = exact erf GELU (configuration_flmr.py:227)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def multi_head_attention(q: Tensor, k: Tensor, v: Tensor, heads: int,
                         add_mask: Optional[Tensor]) -> Tensor:
    """softmax(Q K^T / sqrt(dh) + mask) V  — HF 4.38 BertSelfAttention eager path
    (transformers/models/bert/modeling_bert.py, called from modeling_flmr.py:1622 and
    attention_fusion.py:133-144).  q:[B,Tq,H] k,v:[B,Tk,H]; add_mask broadcastable to
    [B,heads,Tq,Tk] (additive, finfo.min on masked keys)."""
    B, Tq, H = q.shape
    Tk = k.shape[1]
    dh = H // heads
    qh = q.view(B, Tq, heads, dh).transpose(1, 2)
    kh = k.view(B, Tk, heads, dh).transpose(1, 2)
    vh = v.view(B, Tk, heads, dh).transpose(1, 2)
    scores = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if add_mask is not None:
        scores = scores + add_mask
    probs = torch.softmax(scores, dim=-1)
    ctx = probs @ vh
    return ctx.transpose(1, 2).reshape(B, Tq, H)


# ---- device rounding-point emulation (tests only) ------------------------------------------
# The HIP path feeds bf16 operands to the MFMA (fp32 accumulate) and stores Q/K/V and the softmax
# probabilities in bf16; everything else (residual stream, LayerNorm, softmax, GELU) is fp32.  These
# two hooks reproduce exactly those rounding points on top of the fp32 restatement so that parity
# tests can separate "bf16 operand rounding" (inherent, shared with the reference's bf16-mixed runs)
# from kernel bugs.
_RDT = [torch.bfloat16]          # 16-bit operand dtype being emulated (device compute_dtype)
_QSCALE = math.log2(math.e) / 8.0     # log2(e)/sqrt(dh) at dh = 64 (the only head dimension of the path): folded into Wq
_QUERY_NAMES = (".query", ".q_proj")


def _bf(x: Tensor) -> Tensor:
    return x.to(_RDT[-1]).to(torch.float32)


def mm_bf16(x: Tensor, W: Tensor) -> Tensor:
    return _bf(x) @ _bf(W).t()


def multi_head_attention_bf16(q: Tensor, k: Tensor, v: Tensor, heads: int, add_mask: Optional[Tensor]) -> Tensor:
    B, Tq, H = q.shape
    Tk = k.shape[1]
    dh = H // heads
    # device: Q is stored as 16bit(q * log2(e)/sqrt(dh)), so Q K^T is in the log2 domain and the softmax is base 2
    q, k, v = _bf(q * (math.log2(math.e) / math.sqrt(dh))), _bf(k), _bf(v)
    qh = q.view(B, Tq, heads, dh).transpose(1, 2)
    kh = k.view(B, Tk, heads, dh).transpose(1, 2)
    vh = v.view(B, Tk, heads, dh).transpose(1, 2)
    scores = qh @ kh.transpose(-1, -2)
    if add_mask is not None:      # an additive mask (finite in the fusion adjacency) enters the log2 domain scaled as well
        scores = scores + torch.clamp(add_mask * math.log2(math.e), min=FMIN)
    # device: P = exp2(s - max) is rounded to bf16 for the PV MFMA, the row sum stays fp32
    m = scores.max(dim=-1, keepdim=True).values
    e = torch.exp2(scores - m)
    ctx = (_bf(e) @ vh) / e.sum(dim=-1, keepdim=True)
    return ctx.transpose(1, 2).reshape(B, Tq, H)


_MHA = [multi_head_attention]


_FOLD = [False]
_FP8 = [False]
_FP8_DOWN = [False]
_FP8_FIRST = [0]             # "fp8_first_layer": layers below it keep the (folded) 16-bit dataflow
_FP8_QKV = [True]            # "fp8_qkv": e4m3 QKV where the previous layer left e4m3 rows
FP8_GELU_MUL = 8.0          # static scale of the e4m3 GELU output (rr_api.hip FP8_GELU_MUL)


def q8_rows(x: Tensor) -> Tensor:
    """Per-row e4m3 quantise-dequantise (scale = row amax / 448, round to nearest even): the device's layernorm_q8_kernel
    for activations and the weight packer (host_quantize_rows) for output channels."""
    amax = x.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    return (x * (1.0 / s)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * s


def linear_fp8(x: Tensor, w: Dict[str, Tensor], name: str) -> Tensor:
    """y = q8_rows(x) q8_rows(W)^T + b: the e4m3 GEMMs of rr_config.fp8 (exact products, fp32 accumulation)."""
    y = q8_rows(x) @ q8_rows(w[name + ".weight"]).t()
    b = w.get(name + ".bias")
    return y if b is None else y + b


def linear_fp8_static(x: Tensor, w: Dict[str, Tensor], name: str, mul: float = FP8_GELU_MUL) -> Tensor:
    """y = e4m3(clamp(mul x, +-448)) / mul . q8_rows(W)^T + b: the FFN-down of rr_config.fp8 where it runs on the e4m3 ring —
    its A operand is the GELU output under ONE static power-of-two scale (gemm_kernel_hp8 EPI 3), its weights per channel."""
    xq = (x * mul).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * (1.0 / mul)
    y = xq @ q8_rows(w[name + ".weight"]).t()
    b = w.get(name + ".bias")
    return y if b is None else y + b


class device_rounding:
    """`with device_rounding(dtype) as mm:` — run the oracle with the device's 16-bit rounding points
    (dtype = torch.bfloat16 for compute_dtype "bf16", torch.float16 for "fp16").  `fp8`: rr_config.fp8 — in the encoder
    stacks the QKV GEMMs of layers >= 1 and every FFN-up GEMM take per-row e4m3 operands (linear_fp8), the LayerNorm
    kernels are unfolded.  `fold` (default: as the device):
    the encoder stacks (`encoder_stack`) run with LayerNorm folded into the consumer GEMMs, i.e. the 16-bit operand
    of QKV / FFN-up is the RAW pre-LayerNorm row and the normalisation happens on the fp32 accumulators."""

    def __init__(self, dtype=torch.bfloat16, fold: bool = True, fp8: bool = False, fp8_down: bool = False, fp8_first: int = 0,
                 fp8_qkv: bool = True):
        self.dtype, self.fold, self.fp8, self.fp8_down = dtype, fold and not fp8, fp8, fp8 and fp8_down   # fp8_down: FFN-down e4m3 too
        self.fp8_first, self.fp8_qkv = int(fp8_first), bool(fp8_qkv)      # the handle options of the same names (whole stack: 0 / True)

    def __enter__(self):
        _MHA.append(multi_head_attention_bf16)
        _RDT.append(self.dtype)
        _FOLD.append(self.fold)
        _FP8.append(self.fp8)
        _FP8_DOWN.append(self.fp8_down)
        _FP8_FIRST.append(self.fp8_first)
        _FP8_QKV.append(self.fp8_qkv)
        return mm_bf16

    def __exit__(self, *a):
        _MHA.pop()
        _RDT.pop()
        _FOLD.pop()
        _FP8.pop()
        _FP8_DOWN.pop()
        _FP8_FIRST.pop()
        _FP8_QKV.pop()
        return False


def folded_linear(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, W: Tensor, b: Optional[Tensor],
                  wscale: float = 1.0) -> Tensor:
    """Device form of Linear(LayerNorm(x)) with the LayerNorm folded into the GEMM (csrc/gemm_bf16.hip LnResid):
    rstd * (x16 W'^T - mean * c) + d,  W' = 16bit(W * gamma),  c = row sums of W',  d = W beta + b;  mean / variance of the
    fp32 row (the device merges per-128-column (mean, M2) partials: same quantities, fp32)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    Wp = _bf(W * gamma[None, :] * wscale) * (1.0 / wscale)      # wscale: the query weight's log2(e)/sqrt(dh), see linear()
    c = Wp.double().sum(1).float()
    d = (W.double() @ beta.double()).float()
    if b is not None:
        d = d + b
    return rstd * (_bf(x) @ Wp.t() - mu * c) + d


def encoder_stack(h: Tensor, w: Dict[str, Tensor], prefix: str, n_layers: int, heads: int, eps: float,
                  add_mask: Optional[Tensor], mm=None, taps: Optional[dict] = None, tap_name: str = "") -> Tensor:
    """`n_layers` post-LN BertLayers `{prefix}.{i}` on the embedding output `h`.  Plain fp32 (or, with `mm`, the unfolded
    rounding points) = a loop over bert_layer.  Under device_rounding(fold=True) it mirrors the device's folded dataflow:
    layer 0's QKV takes the normalised embedding rows; every FFN-up and every later QKV takes the raw rows of the
    LayerNorm's input with the LayerNorm applied after the GEMM; residuals are the exact fp32 LayerNorm values."""
    if _FP8[-1] and mm is not None:
        # rr_config.fp8 with the handle options "fp8_first_layer" / "fp8_qkv" (rr_api.hip run_layer): layers below `first` run the
        # folded 16-bit dataflow, layers from it on take e4m3 operands for FFN-up and — where the previous layer was an e4m3 layer too
        # and "fp8_qkv" is on — for QKV; the first e4m3 layer behind a folded one reads that layer's raw rows through the folded QKV.
        # ("fp8_first_layer" is an option of the TEXT encoder; the cross-encoder's layers always run the e4m3 configuration)
        first, qkv8 = (_FP8_FIRST[-1] if "context_text_encoder" in prefix else 0), _FP8_QKV[-1]
        pre = g = b = None                      # raw input of the previous (folded) layer's output LayerNorm and its affine
        prev8 = False
        for i in range(n_layers):
            p = f"{prefix}.{i}"
            this8 = i >= first
            if pre is not None:
                q, k, v = (folded_linear(pre, g, b, eps, w[p + f".attention.self.{n}.weight"], w[p + f".attention.self.{n}.bias"],
                                         _QSCALE if n == "query" else 1.0) for n in ("query", "key", "value"))
            elif prev8 and this8 and qkv8:
                q, k, v = (linear_fp8(h, w, p + ".attention.self." + n) for n in ("query", "key", "value"))
            else:
                q, k, v = (linear(h, w, p + ".attention.self." + n, mm) for n in ("query", "key", "value"))
            ctx = _MHA[-1](q, k, v, heads, add_mask)
            if this8:
                a = layer_norm(linear(ctx, w, p + ".attention.output.dense", mm) + h, w, p + ".attention.output.LayerNorm", eps)
                inter = gelu_erf(linear_fp8(a, w, p + ".intermediate.dense"))
                down = linear_fp8_static(inter, w, p + ".output.dense") if _FP8_DOWN[-1] else linear(inter, w, p + ".output.dense", mm)
                h = layer_norm(down + a, w, p + ".output.LayerNorm", eps)
                pre = None
            else:
                pre1 = linear(ctx, w, p + ".attention.output.dense", mm) + h
                g1, b1 = w[p + ".attention.output.LayerNorm.weight"], w[p + ".attention.output.LayerNorm.bias"]
                a = F.layer_norm(pre1, (pre1.shape[-1],), g1, b1, eps)
                inter = gelu_erf(folded_linear(pre1, g1, b1, eps, w[p + ".intermediate.dense.weight"], w[p + ".intermediate.dense.bias"]))
                pre = linear(inter, w, p + ".output.dense", mm) + a
                g, b = w[p + ".output.LayerNorm.weight"], w[p + ".output.LayerNorm.bias"]
                h = F.layer_norm(pre, (pre.shape[-1],), g, b, eps)
            prev8 = this8
            if taps is not None:
                taps[f"{tap_name}{i}"] = h
        return h
    if not (_FOLD[-1] and mm is not None):
        for i in range(n_layers):
            h = bert_layer(h, w, f"{prefix}.{i}", heads, eps, add_mask, mm=mm)
            if taps is not None:
                taps[f"{tap_name}{i}"] = h
        return h
    pre = g = b = None                      # raw input of the previous output LayerNorm and its affine
    for i in range(n_layers):
        p = f"{prefix}.{i}"
        if pre is None:
            q, k, v = (linear(h, w, p + ".attention.self." + n, mm) for n in ("query", "key", "value"))
        else:
            q, k, v = (folded_linear(pre, g, b, eps, w[p + f".attention.self.{n}.weight"], w[p + f".attention.self.{n}.bias"],
                                     _QSCALE if n == "query" else 1.0)
                       for n in ("query", "key", "value"))
        ctx = _MHA[-1](q, k, v, heads, add_mask)
        pre1 = linear(ctx, w, p + ".attention.output.dense", mm) + h
        g1, b1 = w[p + ".attention.output.LayerNorm.weight"], w[p + ".attention.output.LayerNorm.bias"]
        a = F.layer_norm(pre1, (pre1.shape[-1],), g1, b1, eps)                      # fp32 residual value
        inter = gelu_erf(folded_linear(pre1, g1, b1, eps, w[p + ".intermediate.dense.weight"], w[p + ".intermediate.dense.bias"]))
        pre = linear(inter, w, p + ".output.dense", mm) + a
        g, b = w[p + ".output.LayerNorm.weight"], w[p + ".output.LayerNorm.bias"]
        h = F.layer_norm(pre, (pre.shape[-1],), g, b, eps)
        if taps is not None:
            taps[f"{tap_name}{i}"] = h
    return h


def extended_mask(mask01: Tensor) -> Tensor:
    """`get_extended_attention_mask` / `invert_attention_mask` (utils.py:256-282,
    attention_fusion.py:80-82): (1 - m) * finfo(fp32).min, shape [B,1,1,Tk]."""
    return (1.0 - mask01.to(torch.float32))[:, None, None, :] * FMIN


def bert_layer(h: Tensor, w: Dict[str, Tensor], p: str, heads: int, eps: float,
               add_mask: Optional[Tensor], enc_h: Optional[Tensor] = None,
               enc_mask: Optional[Tensor] = None, mm=None) -> Tensor:
    """One post-LN BertLayer; with `enc_h` it is the decoder-configured layer of the
    transformer mapping network: self-attn -> cross-attn -> FFN
    (modeling_flmr.py:640-658, rerank_model.py:450-454)."""
    q = linear(h, w, p + ".attention.self.query", mm)
    k = linear(h, w, p + ".attention.self.key", mm)
    v = linear(h, w, p + ".attention.self.value", mm)
    ctx = _MHA[-1](q, k, v, heads, add_mask)
    a = layer_norm(linear(ctx, w, p + ".attention.output.dense", mm) + h, w,
                   p + ".attention.output.LayerNorm", eps)
    if enc_h is not None:
        q = linear(a, w, p + ".crossattention.self.query", mm)
        k = linear(enc_h, w, p + ".crossattention.self.key", mm)
        v = linear(enc_h, w, p + ".crossattention.self.value", mm)
        ctx = _MHA[-1](q, k, v, heads, enc_mask)
        a = layer_norm(linear(ctx, w, p + ".crossattention.output.dense", mm) + a, w,
                       p + ".crossattention.output.LayerNorm", eps)
    inter = gelu_erf(linear(a, w, p + ".intermediate.dense", mm))
    return layer_norm(linear(inter, w, p + ".output.dense", mm) + a, w, p + ".output.LayerNorm", eps)


def bert_embeddings(w: Dict[str, Tensor], p: str, eps: float, input_ids: Optional[Tensor] = None,
                    token_type_ids: Optional[Tensor] = None,
                    inputs_embeds: Optional[Tensor] = None) -> Tensor:
    """BertEmbeddings: (word | inputs_embeds) + token_type + position -> LayerNorm.
    Absolute positions 0..T-1; token_type defaults to 0 (attention_fusion.py:66-76,126-132)."""
    x = w[p + ".word_embeddings.weight"][input_ids] if inputs_embeds is None else inputs_embeds
    B, T = x.shape[:2]
    if token_type_ids is None:
        token_type_ids = torch.zeros(B, T, dtype=torch.long)
    x = x + w[p + ".token_type_embeddings.weight"][token_type_ids]
    x = x + w[p + ".position_embeddings.weight"][:T][None]
    return layer_norm(x, w, p + ".LayerNorm", eps)


def text_encoder(cfg: OracleConfig, w: Dict[str, Tensor], input_ids: Tensor, attention_mask: Tensor,
                 token_type_ids: Optional[Tensor], mm=None, taps: Optional[dict] = None) -> Tensor:
    """FLMRTextModel.forward -> BertModel (modeling_flmr.py:1637-1688); returns
    last_hidden_state [N,S,H]."""
    p = "context_text_encoder.bert_model"
    h = bert_embeddings(w, p + ".embeddings", cfg.ln_eps, input_ids, token_type_ids)
    if taps is not None:
        taps["text_emb"] = h
    am = extended_mask(attention_mask)
    return encoder_stack(h, w, f"{p}.encoder.layer", cfg.layers, cfg.heads, cfg.ln_eps, am, mm, taps, "text_layer_")


def token_mask(input_ids: Tensor) -> Tensor:
    """`RerankModel.mask` with empty skiplist (rerank_model.py:508-513): id != 0."""
    return (input_ids != 0).to(torch.float32)


def query_stage(cfg: OracleConfig, w: Dict[str, Tensor], input_ids: Tensor, attention_mask: Tensor,
                token_type_ids: Optional[Tensor], image_cls: Optional[Tensor],
                image_patches: Optional[Tensor], mm=None, taps: Optional[dict] = None
                ) -> Tuple[Tensor, Tensor]:
    """`RerankModel.query` (rerank_model.py:333-479) with `mask_instructions=False`.
    The CLIP ViT itself is upstream of the path: `image_cls` is
    `last_hidden_state[:,0]` [N,Vh] (rerank_model.py:411) and `image_patches` is
    `hidden_states[-2][:,1:]` [N,np,Vh] (rerank_model.py:424-426), already repeated
    per pair (rerank_model.py:541-544).
    Returns (late_interaction_output [N,S(+P),D] L2-normalised, query_mask [N,S])."""
    hs = text_encoder(cfg, w, input_ids, attention_mask, token_type_ids, mm, taps)
    text = linear(hs, w, "context_text_encoder_linear", mm)             # :380-382 (no bias)
    mask = token_mask(input_ids)                                         # :385-391
    text = text * mask[..., None]                                        # :392
    Q = text
    if image_cls is not None:
        N = image_cls.shape[0]
        x = linear(image_cls, w, "context_vision_projection.model.0", mm)   # MLP modeling_flmr.py:531-546
        x = linear(torch.tanh(x), w, "context_vision_projection.model.2", mm)
        prefix = x.view(N, -1, cfg.li_dim)                                   # :419-421
        t = linear(image_patches, w, "transformer_mapping_input_linear", mm)  # :428-430
        enc = hs[:, : cfg.cross_attn_len]                                    # :438-442
        enc_mask = extended_mask(torch.ones(N, enc.shape[1]))               # :433-447 (all ones)
        for i in range(cfg.map_layers):
            t = bert_layer(t, w, f"transformer_mapping_network.layer.{i}", cfg.heads, cfg.ln_eps,
                           None, enc_h=enc, enc_mask=enc_mask, mm=mm)       # :450-454
        t = linear(t, w, "transformer_mapping_output_linear", mm)           # :461-465
        Q = torch.cat([text, prefix, t], dim=1)                              # :467-472
        if taps is not None:
            taps["vision_prefix"] = prefix
            taps["vision_mapped"] = t
    Q = F.normalize(Q, p=2, dim=2)                                           # :478 (eps 1e-12)
    if taps is not None:
        taps["late_interaction"] = Q
    return Q, mask


def cross_encoder(cfg: OracleConfig, w: Dict[str, Tensor], inputs_embeds: Tensor, mask01: Tensor,
                  attention_adj: Optional[Tensor] = None, mm=None, taps: Optional[dict] = None
                  ) -> Tuple[Tensor, Tensor]:
    """`CrossEncoder.forward` (utils.py:85-108) over `AttentionFusionBertModel.forward`
    (attention_fusion.py:61-160): embeddings(inputs_embeds) -> Lc layers -> CLS row ->
    classifier1/2.  Pooler output is computed and discarded by the reference; skipped."""
    p = "reranker.bert_model"
    h = bert_embeddings(w, p + ".embeddings", cfg.ln_eps, inputs_embeds=inputs_embeds)
    am = extended_mask(mask01)
    if attention_adj is not None:                                           # attention_fusion.py:84-102
        if attention_adj.dim() == 3:
            attention_adj = attention_adj[:, None]
        am = am + attention_adj
    h = encoder_stack(h, w, f"{p}.encoder.layer", cfg.ce_layers, cfg.ce_heads, cfg.ln_eps, am, mm, taps, "ce_layer_")
    cls = h[:, 0]                                                           # utils.py:102
    return linear(cls, w, "reranker.classifier1"), linear(cls, w, "reranker.classifier2")


# --------------------------------------------------------------------------- heads / loss
def prepare_logits_labels(loss_fn: str, logits: Tensor, logits2: Tensor, Bq: int, num_neg: int,
                          labels: Optional[Sequence[float]]):
    """utils.py:228-254."""
    lab = None
    if labels is not None:
        assert isinstance(labels, list), "Labels must be a list"
        assert loss_fn != "negative_sampling", \
            "Labels should not be provided for negative sampling loss function"
        lab = torch.tensor(labels, dtype=torch.float32).reshape(-1, 1)
    if loss_fn in ("BCE", "2H_BCE"):
        if lab is None:
            lab = torch.zeros(num_neg + 1, 1)
            lab[0, 0] = 1
            lab = lab.repeat(Bq, 1)
        if loss_fn == "2H_BCE":
            lab = lab.view(-1).long()
            logits = torch.cat((logits, logits2), dim=1)
    elif loss_fn == "negative_sampling":
        logits = logits.view(-1, num_neg + 1)
        lab = torch.zeros(Bq, dtype=torch.long)
    else:
        raise ValueError(f"Unknown loss function {loss_fn}")
    return logits, lab


def loss_value(loss_fn: str, pos_weight: Optional[float], logits: Tensor, labels: Tensor) -> Tensor:
    """utils.py:208-224: BCEWithLogits(pos_weight) | CE(weight=[1,pos_weight]) | CE."""
    if loss_fn == "BCE":
        pw = torch.tensor([pos_weight]) if pos_weight is not None else None
        return F.binary_cross_entropy_with_logits(logits, labels, pos_weight=pw)
    if loss_fn == "2H_BCE":
        cw = torch.tensor([1.0, pos_weight]) if pos_weight is not None else None
        return F.cross_entropy(logits, labels, weight=cw)
    if loss_fn == "negative_sampling":
        return F.cross_entropy(logits, labels)
    raise ValueError(f"Unknown loss function {loss_fn}")


@dataclass
class OracleOutput:
    loss: Tensor
    logits: Tensor
    taps: dict = field(default_factory=dict)


def full_context_forward(cfg: OracleConfig, w: Dict[str, Tensor], input_ids: Tensor,
                         attention_mask: Tensor, token_type_ids: Tensor, Bq: int, K: int,
                         image_cls: Optional[Tensor] = None, image_patches: Optional[Tensor] = None,
                         labels: Optional[List[float]] = None, mm=None, want_taps: bool = False
                         ) -> OracleOutput:
    """`FullContextRerankModel.forward` (rerank_model.py:523-591) from the tokenised
    pair batch onward (the tokenizer passes of utils.py:129-167 are upstream).
    `image_cls` [Bq,Vh] / `image_patches` [Bq,np,Vh] are per *query* and repeated per
    pair here exactly as `query_pixel_values.repeat_interleave(K)` does (:541-544)."""
    N = Bq * K
    assert N == input_ids.shape[0]                                          # :527
    if labels:
        assert len(labels) == N                                             # :528-529
    taps = {} if want_taps else None
    if image_cls is not None:
        image_cls = image_cls.repeat_interleave(K, dim=0)
        image_patches = image_patches.repeat_interleave(K, dim=0)
    Q, qmask = query_stage(cfg, w, input_ids, attention_mask, token_type_ids, image_cls,
                           image_patches, mm, taps)
    x = linear(Q, w, "cross_encoder_input_mapping", mm)                     # :557-559
    P = x.shape[1] - qmask.shape[1]
    mask = torch.cat([qmask, torch.ones(N, P)], dim=1) if P > 0 else qmask  # :561-577
    l1, l2 = cross_encoder(cfg, w, x, mask, None, mm, taps)                  # :580-583
    logits, lab = prepare_logits_labels(cfg.loss_fn, l1, l2, Bq, K - 1, labels)  # :584
    loss = loss_value(cfg.loss_fn, cfg.pos_weight, logits, lab)             # :587
    if cfg.loss_fn == "2H_BCE":
        logits = logits[:, 1].unsqueeze(1)                                  # :589-590
    return OracleOutput(loss=loss, logits=logits, taps=taps or {})


def instruction_query_mask(input_ids: Tensor, instruction_token_id: Optional[int]) -> Tensor:
    """`RerankModel.query_mask` (rerank_model.py:481-506): id != 0 and (index > sep or index < 2), sep = first position
    of the instruction token (argmax of an indicator; rows without it give 0, then clamped to 1)."""
    if instruction_token_id is None:
        return token_mask(input_ids)
    sep = torch.argmax((input_ids == instruction_token_id).int(), dim=1)
    sep = torch.clamp(sep, min=1)
    idx = torch.arange(input_ids.shape[1])[None, :]
    return ((input_ids != 0) & ((idx > sep[:, None]) | (idx < 2))).to(torch.float32)


def fusion_adjacency(preflmr_scores: Tensor, ql: int, P: int, S: int, fusion_multiplier: float) -> Tensor:
    """PreFLMR attention fusion (rerank_model.py:276-319): the retriever's raw MaxSim score matrix
    `preflmr_scores` [N, S, ql + P] (context token x query/image token), cut to the context tokens that are in the joint
    sequence, becomes an additive attention bias over the cross-encoder tokens [query | image | context]:
    query->context rows are softmax over the context tokens, context->query rows softmax over the query tokens, the two
    self-attention blocks stay 0; all of it times `fusion_multiplier`."""
    N = preflmr_scores.shape[0]
    ts = preflmr_scores[:, 2: 2 - ql, :]                                                   # :277-279
    assert ts.shape == (N, S - ql, ql + P)                                                  # :280-284
    ul = torch.zeros(N, ql + P, ql + P)
    br = torch.zeros(N, S - ql, S - ql)
    ur = torch.softmax(ts.permute(0, 2, 1), dim=-1)                                         # :305
    bl = torch.softmax(ts, dim=-1)                                                          # :306
    return torch.cat([torch.cat([ul, ur], 2), torch.cat([bl, br], 2)], 1) * fusion_multiplier   # :309-315


def joint_sequence(query_input_ids: Tensor, query_attention_mask: Tensor, context_input_ids: Tensor,
                   context_attention_mask: Tensor, K: int) -> Tuple[Tensor, Tensor]:
    """rerank_model.py:190-224: the query ids repeated K times, followed by the context ids cut to [2, 2 - ql)."""
    ql = query_input_ids.shape[1]
    q_ids = query_input_ids.repeat_interleave(K, 0)
    q_am = query_attention_mask.repeat_interleave(K, 0)
    lt, rt = 2, 2 - ql                                                                      # :204-205
    return (torch.cat([q_ids, context_input_ids[:, lt:rt]], 1),                             # :207-215
            torch.cat([q_am, context_attention_mask[:, lt:rt]], 1))                         # :216-224


def reorder_query_image_context(xin: Tensor, mask01: Tensor, ql: int, S: int) -> Tuple[Tensor, Tensor]:
    """rerank_model.py:241-274: the vision tokens join the mask as ones, then inputs and mask are re-ordered from
    [query | context | image] to [query | image | context]."""
    N, P = xin.shape[0], xin.shape[1] - S
    m = torch.cat([mask01, torch.ones(N, P)], 1)                                            # :245-254
    xin = torch.cat((xin[:, :ql], xin[:, S:], xin[:, ql:S]), 1)                             # :257-264
    m = torch.cat((m[:, :ql], m[:, S:], m[:, ql:S]), 1)                                     # :267-274
    return xin, m


def interaction_fusion_adjacency(preflmr_scores: Tensor, fusion_multiplier: float) -> Tensor:
    """interaction_rerank_model.py:131-142: scores [N, Lc, Lq] -> additive bias over the tokens [query | context]."""
    N, Lc, Lq = preflmr_scores.shape
    ur = torch.softmax(preflmr_scores.permute(0, 2, 1), dim=-1)
    bl = torch.softmax(preflmr_scores, dim=-1)
    return torch.cat([torch.cat([torch.zeros(N, Lq, Lq), ur], 2), torch.cat([bl, torch.zeros(N, Lc, Lc)], 2)], 1) \
        * fusion_multiplier


def rerank_model_forward(cfg: OracleConfig, w: Dict[str, Tensor], query_input_ids: Tensor, query_attention_mask: Tensor,
                         context_input_ids: Tensor, context_attention_mask: Tensor, K: int, image_cls: Tensor,
                         image_patches: Tensor, instruction_token_id: Optional[int] = None, mm=None,
                         want_taps: bool = False, preflmr_scores: Optional[Tensor] = None,
                         fusion_multiplier: float = 1.0) -> OracleOutput:
    """`RerankModel.forward` (rerank_model.py:171-331), with the optional PreFLMR attention fusion."""
    Bq, ql = query_input_ids.shape
    N = Bq * K
    assert N == context_input_ids.shape[0]                                                  # :188
    S = context_input_ids.shape[1]
    assert S == cfg.max_pos                                                                 # :202
    joint_ids, joint_am = joint_sequence(query_input_ids, query_attention_mask, context_input_ids,
                                         context_attention_mask, K)
    taps = {} if want_taps else None
    img_c = image_cls.repeat_interleave(K, 0)
    img_p = image_patches.repeat_interleave(K, 0)
    # query(): same encoder stage, token types default to 0, mask = query_mask(..., mask_instructions)
    hs = text_encoder(cfg, w, joint_ids, joint_am, None, mm, taps)
    text = linear(hs, w, "context_text_encoder_linear", mm)
    mask = instruction_query_mask(joint_ids, instruction_token_id)
    text = text * mask[..., None]
    x = linear(img_c, w, "context_vision_projection.model.0", mm)
    x = linear(torch.tanh(x), w, "context_vision_projection.model.2", mm)
    prefix = x.view(N, -1, cfg.li_dim)
    t = linear(img_p, w, "transformer_mapping_input_linear", mm)
    enc = hs[:, : cfg.cross_attn_len]
    enc_mask = extended_mask(torch.ones(N, enc.shape[1]))
    for i in range(cfg.map_layers):
        t = bert_layer(t, w, f"transformer_mapping_network.layer.{i}", cfg.heads, cfg.ln_eps, None, enc_h=enc,
                       enc_mask=enc_mask, mm=mm)
    t = linear(t, w, "transformer_mapping_output_linear", mm)
    Q = F.normalize(torch.cat([text, prefix, t], 1), p=2, dim=2)
    xin = linear(Q, w, "cross_encoder_input_mapping", mm)                                   # :237-239
    P = xin.shape[1] - S
    xin, m = reorder_query_image_context(xin, mask, ql, S)                                  # :245-274
    adj = None if preflmr_scores is None else fusion_adjacency(preflmr_scores, ql, P, S, fusion_multiplier)
    l1, l2 = cross_encoder(cfg, w, xin, m, adj, mm, taps)                                   # :321-325
    logits, _ = prepare_logits_labels(cfg.loss_fn, l1, l2, Bq, K - 1, None)                 # :327
    # :328 `loss = self.loss_fn(logits, logits)` — the labels are ignored, the logits are their own targets
    if cfg.loss_fn == "BCE":
        pw = torch.tensor([cfg.pos_weight]) if cfg.pos_weight is not None else None
        loss = F.binary_cross_entropy_with_logits(logits, logits, pos_weight=pw)
    elif cfg.loss_fn == "2H_BCE":
        cw = torch.tensor([1.0, cfg.pos_weight]) if cfg.pos_weight is not None else None
        loss = F.cross_entropy(logits, logits, weight=cw)
        logits = logits[:, 1].unsqueeze(1)                                                  # :329-330
    else:
        raise NotImplementedError("RerankModel + negative_sampling is not exercised by any reference config")
    return OracleOutput(loss=loss, logits=logits, taps=taps or {})


def mores_layer(h: Tensor, doc: Tensor, w: Dict[str, Tensor], p: str, heads: int, eps: float, qry_mask: Tensor,
                cross_mask: Tensor, mm=None) -> Tensor:
    """`MORES_BertLayer.forward` (mores_model.py:21-57): cross-attention(query -> doc) FIRST, then self-attention
    over the query tokens, then the FFN; each sub-block is dense + residual + LayerNorm (post-LN)."""
    q = linear(h, w, p + ".crossattention.self.query", mm)
    k = linear(doc, w, p + ".crossattention.self.key", mm)
    v = linear(doc, w, p + ".crossattention.self.value", mm)
    ctx = _MHA[-1](q, k, v, heads, cross_mask)
    a = layer_norm(linear(ctx, w, p + ".crossattention.output.dense", mm) + h, w,
                   p + ".crossattention.output.LayerNorm", eps)
    q = linear(a, w, p + ".attention.self.query", mm)
    k = linear(a, w, p + ".attention.self.key", mm)
    v = linear(a, w, p + ".attention.self.value", mm)
    ctx = _MHA[-1](q, k, v, heads, qry_mask)
    b = layer_norm(linear(ctx, w, p + ".attention.output.dense", mm) + a, w, p + ".attention.output.LayerNorm", eps)
    inter = gelu_erf(linear(b, w, p + ".intermediate.dense", mm))
    return layer_norm(linear(inter, w, p + ".output.dense", mm) + b, w, p + ".output.LayerNorm", eps)


def interaction_forward(cfg: OracleConfig, w: Dict[str, Tensor], query_li: Tensor, context_li: Tensor,
                        query_mask: Tensor, context_mask: Tensor, K: int, labels: Optional[List[float]] = None,
                        mores: bool = False, mm=None, want_taps: bool = False,
                        preflmr_scores: Optional[Tensor] = None, fusion_multiplier: float = 1.0) -> OracleOutput:
    """`InteractionRerankModel.forward` (interaction_rerank_model.py:110-166).
    query_li [Bq,Lq,D], context_li [N,Lc,D], masks 0/1 [Bq,Lq] / [N,Lc]; `preflmr_scores` [N,Lc,Lq] adds the attention
    fusion bias (:131-142; NORMAL interaction type only, MORES raises NotImplementedError, mores_model.py:72-73)."""
    Bq = query_li.shape[0]
    N = Bq * K
    assert N == context_li.shape[0], f"{query_li.shape}, {context_li.shape}, {K - 1}"          # :123
    q = query_li.repeat_interleave(K, dim=0)                                                       # :128
    qm = query_mask.to(torch.float32).repeat_interleave(K, dim=0)                                  # :129
    cm = context_mask.to(torch.float32)
    taps = {} if want_taps else None
    adj = None
    if preflmr_scores is not None:                                                                  # :131-142
        if mores:
            raise NotImplementedError("Attention adj is not implemented for MORES")
        adj = interaction_fusion_adjacency(preflmr_scores, fusion_multiplier)
    if mores:                                                                                       # :147-156
        h = linear(q, w, "cross_encoder_input_mapping", mm)
        doc = linear(context_li.to(torch.float32), w, "cross_encoder_input_mapping", mm)
        qb, cb = extended_mask(qm), extended_mask(cm)                                               # mores_model.py:75-76
        for i in range(cfg.ce_layers):
            h = mores_layer(h, doc, w, f"reranker.interaction_module.{i}", cfg.ce_heads, cfg.ln_eps, qb, cb, mm)
        cls = h[:, 0]                                                                               # mores_model.py:88
        l1, l2 = linear(cls, w, "reranker.classifier1"), linear(cls, w, "reranker.classifier2")
    else:                                                                                           # :157-161
        x = linear(torch.cat((q, context_li), dim=1), w, "cross_encoder_input_mapping", mm)
        l1, l2 = cross_encoder(cfg, w, x, torch.cat((qm, cm), dim=1), adj, mm, taps)
    logits, lab = prepare_logits_labels(cfg.loss_fn, l1, l2, Bq, K - 1, labels)                     # :163
    loss = loss_value(cfg.loss_fn, cfg.pos_weight, logits, lab)                                     # :164
    if cfg.loss_fn == "2H_BCE":      # the reference returns both columns here (:165); ranking uses column 1 everywhere else
        logits = logits[:, 1].unsqueeze(1)
    return OracleOutput(loss=loss, logits=logits, taps=taps or {})


def interaction_weight_spec(cfg: OracleConfig, mores: bool) -> List[Tuple[str, Tuple[int, ...], str]]:
    Hc, Ic, D = cfg.ce_hidden, cfg.ce_intermediate, cfg.li_dim
    s: List[Tuple[str, Tuple[int, ...], str]] = [("cross_encoder_input_mapping.weight", (Hc, D), "w"),
                                                  ("cross_encoder_input_mapping.bias", (Hc,), "b")]
    if mores:
        for i in range(cfg.ce_layers):
            s += _layer_shapes(f"reranker.interaction_module.{i}", Hc, Ic, True)
    else:
        p = "reranker.bert_model"
        s += [(f"{p}.embeddings.position_embeddings.weight", (cfg.ce_max_pos, Hc), "e"),
              (f"{p}.embeddings.token_type_embeddings.weight", (cfg.type_vocab, Hc), "e"),
              (f"{p}.embeddings.LayerNorm.weight", (Hc,), "g"), (f"{p}.embeddings.LayerNorm.bias", (Hc,), "b")]
        for i in range(cfg.ce_layers):
            s += _layer_shapes(f"{p}.encoder.layer.{i}", Hc, Ic, False)
    s += [("reranker.classifier1.weight", (1, Hc), "w"), ("reranker.classifier1.bias", (1,), "b"),
          ("reranker.classifier2.weight", (1, Hc), "w"), ("reranker.classifier2.bias", (1,), "b")]
    return s


def make_interaction_weights(cfg: OracleConfig, mores: bool, seed: int = 0, hf_init: bool = False) -> Dict[str, Tensor]:
    w: Dict[str, Tensor] = {}
    for idx, (name, shape, kind) in enumerate(interaction_weight_spec(cfg, mores)):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if kind in ("w", "e"):
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "g":
            t = torch.ones(shape) if hf_init else 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape) if hf_init else 0.05 * torch.randn(shape, generator=g)
        w[name] = t
    return w


def make_interaction_inputs(cfg: OracleConfig, Bq: int, K: int, Lq: int, Lc: int, seed: int = 2022):
    """Retriever-shaped inputs: unit-norm token embeddings, zero rows where masked (modeling_flmr.py:1365-1370,
    1551-1552), ragged context lengths."""
    g = torch.Generator().manual_seed(seed)
    N = Bq * K
    q = F.normalize(torch.randn(Bq, Lq, cfg.li_dim, generator=g), dim=-1)
    c = F.normalize(torch.randn(N, Lc, cfg.li_dim, generator=g), dim=-1)
    qm = torch.ones(Bq, Lq)
    qm[:, max(2, Lq // 3): max(3, Lq // 2)] = 0                     # some masked query tokens in the middle
    cm = torch.zeros(N, Lc)
    for n in range(N):
        L = int(torch.randint(max(2, Lc // 4), Lc + 1, (1,), generator=g))
        cm[n, :L] = 1
    return q * qm[..., None], c * cm[..., None], qm, cm


# --------------------------------------------------------------------------- rank + metric
def rank_descending_stable(scores: Sequence[float]) -> List[int]:
    """`sorted(zip(docs, logits), key=score, reverse=True)`
    (Reranker_base_executor.py:934-935).  Python's sort is stable and `reverse=True`
    preserves the original order of equal keys, so ties keep retrieval order."""
    return [i for i, _ in sorted(enumerate(scores), key=lambda t: t[1], reverse=True)]


def recall_precision_at_k(ranked_ids: Sequence[Sequence], pos_ids: Sequence[Sequence],
                          Ks: Sequence[int]) -> Dict[str, List[float]]:
    """`compute_rerank_DPR_scores_with_pos_ids` (metrics_processors.py:816-890):
    recall@K = mean over queries of 1[any positive in top-K]; precision@K = mean of
    hits-in-top-K / K."""
    rec = [0.0] * len(Ks)
    prec = [0.0] * len(Ks)
    for ids, pos in zip(ranked_ids, pos_ids):
        hit = [1 if pid in pos else 0 for pid in ids[: max(Ks)]]
        for j, k in enumerate(Ks):
            s = sum(hit[:k])
            rec[j] += 1.0 if s > 0 else 0.0
            prec[j] += s / k
    n = max(1, len(ranked_ids))
    return {"recall": [r / n for r in rec], "precision": [p / n for p in prec]}


# --------------------------------------------------------------------------- CLIP vision tower
VIT_PREFIX = "context_vision_encoder.vision_model.vision_model"   # FLMRVisionModel.vision_model = CLIPVisionModel
VIT_LN_EPS = 1e-5                                                 # CLIPVisionConfig.layer_norm_eps


def quick_gelu(x: Tensor) -> Tensor:
    """CLIP's hidden_act "quick_gelu": x * sigmoid(1.702 x) (transformers activations.QuickGELUActivation)."""
    return x * torch.sigmoid(1.702 * x)


def clip_vision_forward(cfg: OracleConfig, w: Dict[str, Tensor], pixel_values: Tensor, mm=None
                        ) -> Tuple[Tensor, Tensor]:
    """FLMRVisionModel.forward(pixel_values, output_hidden_states=True) (modeling_flmr.py:1701-1757) =
    HF 4.38 CLIPVisionTransformer: Conv2d(3, Vh, kernel = stride = patch, bias=False) -> [class | patches] +
    position_embedding -> pre_layrnorm -> pre-LN encoder layers (q scaled by dh^-0.5 after q_proj, quick-GELU MLP).
    Returns what the rerankers take from it (rerank_model.py:408-411,424-426): last_hidden_state[:, 0]
    (post_layernorm only touches pooler_output, which is not used) and hidden_states[-2][:, 1:]."""
    v, ps, heads = VIT_PREFIX, cfg.vit_patch_size, cfg.vit_heads
    B = pixel_values.shape[0]
    Vh = cfg.vision_hidden
    cols = F.unfold(pixel_values.to(torch.float32), kernel_size=ps, stride=ps).transpose(1, 2)   # [B, np, 3*ps*ps] (c,ky,kx)
    Wp = w[f"{v}.embeddings.patch_embedding.weight"].reshape(Vh, -1)
    patches = mm(cols, Wp) if mm is not None else cols @ Wp.t()
    x = torch.cat([w[f"{v}.embeddings.class_embedding"].expand(B, 1, Vh), patches], dim=1)
    x = x + w[f"{v}.embeddings.position_embedding.weight"][None, : x.shape[1]]
    x = layer_norm(x, w, f"{v}.pre_layrnorm", VIT_LN_EPS)
    states = [x]
    for i in range(cfg.vit_layers):
        l = f"{v}.encoder.layers.{i}"
        h = layer_norm(x, w, f"{l}.layer_norm1", VIT_LN_EPS)
        q = linear(h, w, f"{l}.self_attn.q_proj", mm)
        k = linear(h, w, f"{l}.self_attn.k_proj", mm)
        vv = linear(h, w, f"{l}.self_attn.v_proj", mm)
        ctx = _MHA[-1](q, k, vv, heads, None)          # (q * dh^-0.5) k^T == q k^T / sqrt(dh)
        x = x + linear(ctx, w, f"{l}.self_attn.out_proj", mm)
        h = layer_norm(x, w, f"{l}.layer_norm2", VIT_LN_EPS)
        x = x + linear(quick_gelu(linear(h, w, f"{l}.mlp.fc1", mm)), w, f"{l}.mlp.fc2", mm)
        states.append(x)
    return states[-1][:, 0], states[-2][:, 1:]


def vit_weight_spec(cfg: OracleConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    Vh, Iv, ps, v = cfg.vision_hidden, cfg.vit_intermediate, cfg.vit_patch_size, VIT_PREFIX
    s: List[Tuple[str, Tuple[int, ...], str]] = [
        (f"{v}.embeddings.class_embedding", (Vh,), "e"),
        (f"{v}.embeddings.patch_embedding.weight", (Vh, 3, ps, ps), "w"),
        (f"{v}.embeddings.position_embedding.weight", (cfg.n_patches + 1, Vh), "e"),
        (f"{v}.pre_layrnorm.weight", (Vh,), "g"), (f"{v}.pre_layrnorm.bias", (Vh,), "b")]
    for i in range(cfg.vit_layers):
        l = f"{v}.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += [(f"{l}.self_attn.{n}.weight", (Vh, Vh), "w"), (f"{l}.self_attn.{n}.bias", (Vh,), "b")]
        s += [(f"{l}.layer_norm1.weight", (Vh,), "g"), (f"{l}.layer_norm1.bias", (Vh,), "b"),
              (f"{l}.mlp.fc1.weight", (Iv, Vh), "w"), (f"{l}.mlp.fc1.bias", (Iv,), "b"),
              (f"{l}.mlp.fc2.weight", (Vh, Iv), "w"), (f"{l}.mlp.fc2.bias", (Vh,), "b"),
              (f"{l}.layer_norm2.weight", (Vh,), "g"), (f"{l}.layer_norm2.bias", (Vh,), "b")]
    return s


def make_vit_weights(cfg: OracleConfig, seed: int = 0, hf_init: bool = False) -> Dict[str, Tensor]:
    """Seeded synthetic CLIP-tower weights (same per-tensor generator scheme as make_weights, offset so the
    two sets never share a stream)."""
    w: Dict[str, Tensor] = {}
    for idx, (name, shape, kind) in enumerate(vit_weight_spec(cfg)):
        g = torch.Generator().manual_seed(seed * 1000003 + 500000 + idx)
        if kind in ("w", "e"):
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "g":
            t = torch.ones(shape) if hf_init else 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape) if hf_init else 0.05 * torch.randn(shape, generator=g)
        w[name] = t
    return w


def make_pixel_values(cfg: OracleConfig, B: int, seed: int = 2022) -> Tensor:
    """CLIP-normalised pixels are roughly N(0, 1.2) per channel; seeded stand-in."""
    g = torch.Generator().manual_seed(seed + 11)
    return 1.2 * torch.randn(B, 3, cfg.vit_image_size, cfg.vit_image_size, generator=g)


# --------------------------------------------------------------------------- synthetic data
WEIGHT_SPEC_DOC = "see make_weights"


def _layer_shapes(p: str, H: int, I: int, cross: bool) -> List[Tuple[str, Tuple[int, ...], str]]:
    out = []
    atts = ["attention"] + (["crossattention"] if cross else [])
    for a in atts:
        for n in ("query", "key", "value"):
            out += [(f"{p}.{a}.self.{n}.weight", (H, H), "w"), (f"{p}.{a}.self.{n}.bias", (H,), "b")]
        out += [(f"{p}.{a}.output.dense.weight", (H, H), "w"), (f"{p}.{a}.output.dense.bias", (H,), "b"),
                (f"{p}.{a}.output.LayerNorm.weight", (H,), "g"), (f"{p}.{a}.output.LayerNorm.bias", (H,), "b")]
    out += [(f"{p}.intermediate.dense.weight", (I, H), "w"), (f"{p}.intermediate.dense.bias", (I,), "b"),
            (f"{p}.output.dense.weight", (H, I), "w"), (f"{p}.output.dense.bias", (H,), "b"),
            (f"{p}.output.LayerNorm.weight", (H,), "g"), (f"{p}.output.LayerNorm.bias", (H,), "b")]
    return out


def weight_spec(cfg: OracleConfig, vision: bool) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (name, shape, kind) list of every tensor the path reads; kind in
    {w: matrix, b: bias/beta, g: LN gamma, e: embedding}."""
    H, I, D = cfg.hidden, cfg.intermediate, cfg.li_dim
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    p = "context_text_encoder.bert_model"
    s += [(f"{p}.embeddings.word_embeddings.weight", (cfg.vocab_size, H), "e"),
          (f"{p}.embeddings.position_embeddings.weight", (cfg.max_pos, H), "e"),
          (f"{p}.embeddings.token_type_embeddings.weight", (cfg.type_vocab, H), "e"),
          (f"{p}.embeddings.LayerNorm.weight", (H,), "g"), (f"{p}.embeddings.LayerNorm.bias", (H,), "b")]
    for i in range(cfg.layers):
        s += _layer_shapes(f"{p}.encoder.layer.{i}", H, I, False)
    s += [("context_text_encoder_linear.weight", (D, H), "w")]
    if vision:
        Vh, PL = cfg.vision_hidden, cfg.prefix_len
        s += [("context_vision_projection.model.0.weight", (D * PL // 2, Vh), "w"),
              ("context_vision_projection.model.0.bias", (D * PL // 2,), "b"),
              ("context_vision_projection.model.2.weight", (D * PL, D * PL // 2), "w"),
              ("context_vision_projection.model.2.bias", (D * PL,), "b"),
              ("transformer_mapping_input_linear.weight", (H, Vh), "w"),
              ("transformer_mapping_input_linear.bias", (H,), "b")]
        for i in range(cfg.map_layers):
            s += _layer_shapes(f"transformer_mapping_network.layer.{i}", H, I, True)
        s += [("transformer_mapping_output_linear.weight", (D, H), "w"),
              ("transformer_mapping_output_linear.bias", (D,), "b")]
    Hc, Ic = cfg.ce_hidden, cfg.ce_intermediate
    s += [("cross_encoder_input_mapping.weight", (Hc, D), "w"), ("cross_encoder_input_mapping.bias", (Hc,), "b")]
    p = "reranker.bert_model"
    s += [(f"{p}.embeddings.position_embeddings.weight", (cfg.ce_max_pos, Hc), "e"),
          (f"{p}.embeddings.token_type_embeddings.weight", (cfg.type_vocab, Hc), "e"),
          (f"{p}.embeddings.LayerNorm.weight", (Hc,), "g"), (f"{p}.embeddings.LayerNorm.bias", (Hc,), "b")]
    for i in range(cfg.ce_layers):
        s += _layer_shapes(f"{p}.encoder.layer.{i}", Hc, Ic, False)
    s += [("reranker.classifier1.weight", (1, Hc), "w"), ("reranker.classifier1.bias", (1,), "b"),
          ("reranker.classifier2.weight", (1, Hc), "w"), ("reranker.classifier2.bias", (1,), "b")]
    return s


def make_weights(cfg: OracleConfig, seed: int = 0, vision: bool = False, hf_init: bool = False,
                 gain: float = 1.0) -> Dict[str, Tensor]:
    """Seeded synthetic weights.  `hf_init=True`: HF init (matrices/embeddings
    N(0,0.02), LN gamma 1 / beta 0, biases 0 — modeling_flmr.py:199-214) as the bench
    uses; default (tests): same matrices but non-trivial biases/gamma/beta so that
    every epilogue term is exercised.  Each tensor draws from its own generator
    seeded by (seed, index) so the result does not depend on which tensors are present."""
    w: Dict[str, Tensor] = {}
    for idx, (name, shape, kind) in enumerate(weight_spec(cfg, vision)):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if kind in ("w", "e"):
            # `gain` > 1 widens the Linear matrices only (not the embeddings): a 0.02-std random network is nearly linear
            # and maps every candidate of a query to almost the same CLS state (logit spread 0.007 at bert-base), so
            # rankings are noise; at gain 2.5 attention is peaked and the spread is ~0.2 (tests/golden c3_sep)
            t = torch.randn(shape, generator=g) * (0.02 * (gain if kind == "w" else 1.0))
        elif kind == "g":
            t = torch.ones(shape) if hf_init else 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape) if hf_init else 0.05 * torch.randn(shape, generator=g)
        w[name] = t
    return w


def make_pair_batch(cfg: OracleConfig, Bq: int, K: int, S: int, seed: int = 2022,
                    regime: str = "realistic", q_len: int = 32):
    """Synthetic tokenised pair batch per SURVEY §8d: [CLS] q.. [SEP] ctx.. [SEP] pad*;
    body ids uniform in [1000, vocab); token_type 0 on `[CLS] q [SEP]`, 1 on `ctx [SEP]`,
    0 on padding; `regime="full"`: all S positions real, `"realistic"`: total length
    ~ U[min(64,S//2), S].  Returns int64 (ids, attention_mask, token_type_ids)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    N = Bq * K
    ids = np.zeros((N, S), dtype=np.int64)
    tt = np.zeros((N, S), dtype=np.int64)
    lo = min(1000, cfg.vocab_size // 2)
    ql = min(q_len, max(1, S // 4))
    for n in range(N):
        L = S if regime == "full" else int(rng.integers(min(64, S // 2), S + 1))
        L = max(L, ql + 4)
        body = rng.integers(lo, cfg.vocab_size, size=L)
        body[0] = 101
        body[ql + 1] = 102
        body[L - 1] = 102
        ids[n, :L] = body
        tt[n, ql + 2: L] = 1
    am = (ids != 0).astype(np.int64)
    return torch.from_numpy(ids), torch.from_numpy(am), torch.from_numpy(tt)


def make_image_feats(cfg: OracleConfig, Bq: int, seed: int = 2022):
    g = torch.Generator().manual_seed(seed + 7)
    cls = torch.randn(Bq, cfg.vision_hidden, generator=g)
    patches = torch.randn(Bq, cfg.n_patches, cfg.vision_hidden, generator=g)
    return cls, patches


def flops_per_pair(cfg: OracleConfig, S: int, vision: bool) -> float:
    """SURVEY §8d algorithmic FLOPs per pair: F_layer(T) = 8TH^2 + 4T^2H + 4THI;
    F_pair = L F_layer(S) + 2SHD + 2(S+P)DHc + Lc F_layer_c(S+P) + F_vis."""
    def fl(T, H, I):
        return 8.0 * T * H * H + 4.0 * T * T * H + 4.0 * T * H * I
    H, I, D = cfg.hidden, cfg.intermediate, cfg.li_dim
    P = (cfg.prefix_len + cfg.n_patches) if vision else 0
    f = cfg.layers * fl(S, H, I) + 2.0 * S * H * D + 2.0 * (S + P) * D * cfg.ce_hidden \
        + cfg.ce_layers * fl(S + P, cfg.ce_hidden, cfg.ce_intermediate)
    if vision:
        Vh, npat, ca = cfg.vision_hidden, cfg.n_patches, cfg.cross_attn_len
        mid = D * cfg.prefix_len // 2
        f += 2.0 * (Vh * mid + mid * D * cfg.prefix_len)                       # MLP
        f += 2.0 * npat * Vh * H + 2.0 * npat * H * D                            # in/out linears
        per = (8.0 * npat * H * H + 4.0 * npat * npat * H                        # self-attn
               + 4.0 * npat * H * H + 4.0 * ca * H * H + 4.0 * npat * ca * H     # cross-attn (q,o | k,v | scores)
               + 4.0 * npat * H * I)
        f += cfg.map_layers * per
    return f
