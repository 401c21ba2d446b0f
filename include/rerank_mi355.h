/*
 * rerank_mi355.h — C ABI of librerank_mi355.so
 *
 * MI355X (gfx950) native cross-encoder rerank forward: the batched
 * (query x K-candidate) BERT-style encoder, the 128-d late-interaction bottleneck,
 * the optional vision prefix / mapping network, the Lc-layer cross encoder, the
 * pointwise (sigmoid/BCE), two-head (softmax/CE) or listwise (softmax/CE) scoring head
 * and the descending stable top-K order, executed as hand-written HIP kernels on a
 * caller-supplied HIP stream.
 *
 * Drop-in boundary (citations are into /root/reference/):
 *   - the call `self.reranker(**batch_input)` in the Rerank executor
 *     (src/executors/Reranker_base_executor.py:603,758,902-903,922) whose callee is
 *     FullContextRerankModel.forward (src/models/rerank/rerank_model.py:523-591) /
 *     RerankModel.forward (:171-331);
 *   - the reference has no FFI of its own for this path (it is pure PyTorch), so the
 *     entry points below are what a ctypes/pybind binding of that module's
 *     __init__/load_state_dict/forward would bind.  INTEGRATION.md shows the
 *     reference-side stub.
 *
 * Conventions
 *   - plain C, no torch types; every pointer marked DEVICE is a HIP device pointer
 *     that the library BORROWS for the duration of the call; HOST pointers are read
 *     before the call returns.
 *   - every function returns rr_status (0 = ok, <0 = error); rr_last_error() gives a
 *     human-readable message for the last failure on that handle.  The library never
 *     aborts/exits and never silently falls back to a CPU path.
 *   - a handle is NOT thread-safe; several handles per process are fine.  All kernels
 *     are enqueued on the hipStream_t passed in (as void*); rr_forward does not
 *     synchronise the device or the stream.
 */
#ifndef RERANK_MI355_H
#define RERANK_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 2

typedef enum rr_status {
  RR_OK = 0,
  RR_ERR_BAD_ARG = -1,      /* null pointer / bad enum / handle misuse            (Python: ValueError)          */
  RR_ERR_BAD_SHAPE = -2,    /* N != Bq*K, S > max_pos, label count ...            (Python: AssertionError,
                                rerank_model.py:188-190,202,527-529)                                             */
  RR_ERR_BAD_DTYPE = -3,    /* unsupported weight dtype                            (Python: ValueError,
                                attention_fusion.py:97-100)                                                      */
  RR_ERR_UNSUPPORTED = -4,  /* configuration the kernels do not cover              (Python: NotImplementedError,
                                rerank_model.py:185, mores_model.py:72-73)                                       */
  RR_ERR_HIP = -5,          /* a HIP runtime call failed                           (Python: RuntimeError)        */
  RR_ERR_OOM = -6,          /* device allocation failed                            (Python: MemoryError)         */
  RR_ERR_MISSING_WEIGHT = -7, /* rr_finalize_weights: a required tensor was never loaded (Python: KeyError)      */
  RR_ERR_NO_DEVICE = -8,    /* no gfx950 device visible                            (Python: RuntimeError)        */
  RR_ERR_RANGE = -9         /* an EARLIER forward of this handle raised the fp16 activation range flag: its logits are
                               unreliable (rr_activation_range_flag below; no reference counterpart) (Python: OverflowError) */
} rr_status;

typedef enum rr_dtype { RR_F32 = 0, RR_BF16 = 1, RR_F16 = 2 } rr_dtype;

/* scoring head / loss (src/models/rerank/utils.py:208-254) */
typedef enum rr_loss_kind {
  RR_LOSS_BCE = 0,               /* pointwise: BCEWithLogits(pos_weight), logits [N,1]                */
  RR_LOSS_2H_BCE = 1,            /* two heads: CrossEntropy([l1,l2], weight=[1,pos_weight]); logits_out = l2 */
  RR_LOSS_NEGATIVE_SAMPLING = 2  /* listwise: logits viewed [Bq,K], CrossEntropy(target = 0)          */
} rr_loss_kind;

/* Architecture.  Field names follow the reference configs
 * (configuration_flmr.py:220-236,332-350; monoBERT_pointwise.jsonnet:111-122). */
typedef struct rr_config {
  int32_t abi_version;        /* = RR_ABI_VERSION */
  /* FLMR text encoder (= BertModel) */
  int32_t vocab_size, hidden, layers, heads, intermediate, max_pos, type_vocab;
  float ln_eps;
  int32_t li_dim;             /* late-interaction dim (128) */
  /* cross encoder (rerank_model.py:89-101) */
  int32_t ce_hidden, ce_layers, ce_heads, ce_intermediate, ce_max_pos;
  /* vision prefix + transformer mapping network (modeling_flmr.py:603-664); has_vision=0 => text_only */
  int32_t has_vision, vision_hidden, prefix_len, n_patches, map_layers, cross_attn_len;
  /* head */
  int32_t loss_kind;          /* rr_loss_kind */
  float pos_weight;           /* NaN = none (utils.py:210,215) */
  int32_t device;             /* HIP device ordinal */
  int32_t compute_dtype;      /* 16-bit MFMA operand type: 0 = bf16 (default; the reference runs bf16-mixed),
                                 1 = fp16 (same MFMA rate, 3 more mantissa bits: logits within 1e-3 of the fp32
                                 forward; range +-65504 is ample for BERT activations, accumulation/residual/
                                 LayerNorm/softmax stay fp32 in both modes) */
  int32_t model_kind;         /* rr_model_kind: which reference reranker class the handle stands for */
  /* optional CLIP ViT image-feature producer (rr_encode_image); vit_layers = 0 => features come from the caller.
     Defaults of FLMRVisionConfig (configuration_flmr.py:90-104): 12 layers, 12 heads, 3072, 224, 32; hidden =
     vision_hidden, n_patches must equal (vit_image_size / vit_patch_size)^2, head dim 64. */
  int32_t vit_layers, vit_heads, vit_intermediate, vit_image_size, vit_patch_size;
  int32_t fp8;                /* 1 = BASELINE configs[4] mode: the QKV (layers >= 1 of a stack) and FFN-up GEMMs of the text and
                                 cross encoder take e4m3 operands on the block-scaled matrix core — activations quantised per row
                                 by the LayerNorm that produces them, weights per output channel at rr_finalize_weights —
                                 while attention and the GEMMs feeding the residual stream stay in compute_dtype.  Logit drift
                                 is reported separately (north_star's 1e-3 is a 16-bit figure); needs hidden % 128 == 0. */
} rr_config;

/* reranker families */
typedef enum rr_model_kind {
  RR_MODEL_FULL_CONTEXT = 0,  /* FullContextRerankModel (monoBERT / monoPreFLMR), rerank_model.py:515-591  -> rr_forward */
  RR_MODEL_INTERACTION = 1,   /* InteractionRerankModel, interaction_type NORMAL (ModPreFLMR-BERT),
                                 interaction_rerank_model.py:110-166                          -> rr_forward_interaction */
  RR_MODEL_MORES = 2          /* InteractionRerankModel, interaction_type MORES (ModPreFLMR-IB), mores_model.py:21-94 */
} rr_model_kind;

typedef struct rr_model* rr_handle;

/* Per-kernel-class device time accumulated while profiling is on (rr_set_profiling). */
typedef enum rr_kernel_class {
  RR_K_GEMM = 0, RR_K_ATTENTION = 1, RR_K_LAYERNORM = 2, RR_K_EMBED = 3, RR_K_TAIL = 4, RR_K_HEAD = 5,
  RR_K_GEMM_FP8 = 6,          /* the e4m3 GEMM launches of rr_config.fp8 (priced against the fp8 matrix-core peak) */
  RR_K_COUNT = 7
} rr_kernel_class;

typedef struct rr_profile {
  double ms[RR_K_COUNT];      /* summed device ms per class (HIP events on the work stream) */
  int64_t launches[RR_K_COUNT];
  double flops[RR_K_COUNT];   /* algorithmic FLOPs issued per class (2*M*N*K, 4*T^2*dh per head ...) */
  double bytes[RR_K_COUNT];   /* algorithmic HBM bytes (operands read once + results written once) */
} rr_profile;

const char* rr_version(void);
const char* rr_status_string(int status);

/* rr_create: allocate a model for `cfg` on cfg->device.  Replaces RerankerClass(reranker_config)
 * (Reranker_base_executor.py:191-202 -> rerank_model.py:81-101,515-520). */
int rr_create(const rr_config* cfg, rr_handle* out);
int rr_destroy(rr_handle h);
const char* rr_last_error(rr_handle h);

/* rr_load_weight: copy ONE tensor (HOST pointer, row-major, `dtype`) into library-owned
 * device memory, converting/packing as the kernels need.  `name` is the reference
 * state_dict key without the executor's `reranker.` prefix, e.g.
 * "context_text_encoder.bert_model.encoder.layer.0.attention.self.query.weight"
 * (Reranker_base_executor.py:351-381 loads these with strict=False: unknown names are
 * ignored and reported through *known = 0).  Replaces load_state_dict. */
int rr_load_weight(rr_handle h, const char* name, const void* host_data, int dtype,
                   int ndim, const int64_t* shape, int* known);
/* rr_finalize_weights: verify every tensor the configured path reads was loaded, build
 * fused/packed forms (QKV concat, log2(e)/sqrt(dh) folded into Wq and bq).  Must precede rr_forward. */
int rr_finalize_weights(rr_handle h);
/* number of tensors the configured path requires, and the i-th required name */
int rr_num_required_weights(rr_handle h);
const char* rr_required_weight_name(rr_handle h, int i);

/* Bytes of grow-only device workspace rr_forward would hold for this shape. */
int64_t rr_workspace_bytes(rr_handle h, int n_pairs, int seq_len);

/* rr_reserve: allocate once, outside the forward, everything a forward over at most n_pairs pairs of n_queries queries
 * would otherwise grow on first use (workspace, the attention kernels' redo flags for `hip_stream`, and with
 * with_fusion != 0 the attention-fusion bias).  len_a = seq_len for full-context models, Lq for interaction models;
 * len_b = Lc (interaction models only).  After rr_reserve no forward of that or a smaller shape allocates, frees or
 * synchronises, which makes the forward capturable into a hipGraph; without it the first forward of a larger shape
 * grows the buffers (synchronising the stream) and a forward under stream capture that would have to grow fails with
 * RR_ERR_BAD_ARG.  Graphs captured earlier stay replayable when a LATER call on the same handle (a larger rr_forward,
 * rr_encode_image, the fusion bias) outgrows a buffer: once rr_reserve has been called or a capture has been seen on the
 * handle, an outgrown block is retired until rr_destroy instead of freed, and the attention redo-flag buffers are never
 * freed; such a replay computes in the old blocks it was captured with (and overlaps nothing the handle still uses).  What
 * a replay does NOT survive: rr_destroy of the handle.  The module's constructor-time analogue in the reference: none
 * (PyTorch's caching allocator). */
int rr_reserve(rr_handle h, int n_pairs, int n_queries, int len_a, int len_b, int with_fusion, void* hip_stream);

/* Length-bucketed execution (SURVEY.md "Variable length"; the reference pads every pair to max_decoder_source_length,
 * utils.py:157-165).  After rr_set_padded_seq_len(h, S_pad) a full-context rr_forward may be called with a seq_len S <= S_pad —
 * the host hands over only the first S columns of the pairs whose real length fits — and computes what the S_pad-long call
 * computes for those pairs: text positions are 0..S-1 either way, padded keys contribute exactly 0 to every softmax, and the
 * vision tokens of the cross-encoder keep the positions S_pad.. they have behind the padded text.  Text-only models: logits bit
 * for bit those of the padded call; with vision tokens the cross-encoder's key tiles are cut at other places, i.e. equal up to
 * fp32 summation order.  "Bit for bit" holds while the shorter call's attention grid and the padded call's are on the same
 * side of the 1 024-workgroup threshold between the online and the fixed-reference softmax schedule (a bucket of a few
 * pairs runs the online form: same values up to rounding); rr_forward_packed has no such condition.  0 switches it off.
 * ROW-COUNT DEPENDENCE OF THE ROUNDINGS (every forward entry point): a GEMM row's values never depend on the rows that share its
 * launch, but the form the residual stream takes between the two residual epilogues of a layer does — (hi, lo) 16-bit pairs where
 * the call has at least 128 tiles of 256 x 256 rows x hidden (about 11k rows at hidden 768: the persistent ring runs there),
 * fp32 rows below.  A bucket, a pair_begin / pair_end slice or a multi-GPU shard under that size therefore agrees with the
 * large call to the parity tolerance (1e-3 in fp16; measured ~2e-4 on c3_full: tests/test_gpu_parity_fullsize.py,
 * test_sharded_slices_and_the_whole_list_agree_within_the_parity_gate), not to the bit.  "resid_split" = 0 removes the
 * dependence (fp32 rows at every size, +3 % time at the bench shape).
 * Python: RerankEngine.forward_ids_bucketed. */
int rr_set_padded_seq_len(rr_handle h, int padded_seq_len);

/* Per-handle numerics options: everything that changes WHICH arithmetic a forward of this handle runs lives in the handle
 * ("no global state => several handles per process"; the reference's analogue is the reranker_config each
 * module instance owns, Reranker_base_executor.py:191-202).  Two handles of one process may differ in any of them; a change
 * takes effect with the next forward of THAT handle.
 *   key              values                 default   meaning
 *   "ln_lite"        0 | 1                  1         1: residuals are recomputed from LayerNorm statistics; 0: every LayerNorm
 *                                                     writes its fp32 output and the residual GEMMs read it back (reference dataflow)
 *   "ln_fold"        0 | 1                  1         1: LayerNorm folded into the consumer GEMMs (QKV, FFN-up); 0: LayerNorm kernels
 *   "resid_split"    0 | 1                  1         1: pre-LayerNorm rows between the residual epilogues as 16-bit hi + lo
 *                                                     (where the persistent ring runs); 0: fp32 rows
 *   "resid_lo8"      0 | 1                  by dtype  form of that lo half.  1: e5m2 bytes of (x - hi) * 16 (6 bytes per element through
 *                                                     the residual epilogues instead of 8: -1.6 % step time at the bench shape; x kept to
 *                                                     >= 14 significant bits behind an fp16 hi, 11 behind a bf16 hi); 0: fp16 (22 / 19
 *                                                     bits).  Default: 1 for fp16 handles (drift against the fp32 goldens unchanged),
 *                                                     0 for bf16 handles (it would add ~20 % to theirs)
 *   "ce_cls_only"    0 | 1                  1         1: the cross-encoder's last layer computes queries / FFN for the CLS row of a
 *                                                     pair only (all the classifiers read, utils.py:105-108); 0: every row
 *   "fp8_ffn_down"   0 | 1                  0         rr_config.fp8 only: FFN-down on the e4m3 ring too (GELU output as e4m3 under a
 *                                                     static scale of 8); 0: FFN-down keeps 16-bit operands
 *   "fp8_first_layer" 0 .. layers           layers-1  rr_config.fp8 only: text-encoder layers with an index below this value keep
 *                                                     16-bit operands (the folded-LayerNorm dataflow); layers from it on run the
 *                                                     e4m3 configuration.  `layers` (or more) = no e4m3 GEMM at all; 0 = every
 *                                                     layer (the whole-stack form: 1.13x the 16-bit line on bert-large, and
 *                                                     MEASURED not to keep the fp32 top-5 on the ranking fixtures — opt in only
 *                                                     with a checkpoint you have validated).  A perturbation injected early is
 *                                                     amplified by every later layer, so e4m3 goes into the LAST layers first;
 *                                                     the default (the last layer only) is the largest subset that ranks with a margin that survives
 *                                                     a re-draw of unrelated roundings (DESIGN.md "fp8")
 *   "fp8_qkv"        0 | 1                  1         rr_config.fp8 only: 0 = of an e4m3 layer only the FFN takes e4m3 operands,
 *                                                     its QKV projection keeps 16-bit ones
 *   "attn_fixed_ref" 0 | 1 | 2 | 3          3         softmax schedule of large attention grids: 0 online only, 1 fixed reference
 *                                                     with 32 query rows per wave, 2 with 64, 3: 2 where 256-row workgroups pad no
 *                                                     more rows than 128-row ones, else 1
 * value -1 = "not set": the handle follows the process-wide diagnostic switch of the same name (rr_set_tuning), which is what
 * every option starts as.  rr_get_option returns the EFFECTIVE value.  RR_ERR_BAD_ARG: unknown key, value out of range. */
int rr_set_option(rr_handle h, const char* key, int value);
int rr_get_option(rr_handle h, const char* key, int* value_out);

/* Range guard of the 16-bit residual rows.  With the folded LayerNorm the RAW pre-LayerNorm rows are MFMA operands and the
 * `hi` half of the residual stream; fp16 (the mode that meets 1e-3) ends at 65 504.  The kernel that merges the rows'
 * LayerNorm statistics raises a device-side flag when a row's sum of squares reaches 9e8 (no element of a row below that can
 * exceed 3e4) or is not finite; BERT-family activations stay four orders of magnitude below.  rr_activation_range_flag
 * copies the flag to *flag_out (synchronises `hip_stream`; call it outside the hot loop, e.g. once per evaluation batch
 * group) and clears it when reset != 0.  A raised flag means: rebuild the handle with compute_dtype = 0 (bf16, the
 * reference's autocast type, same exponent range as fp32).  Python: RerankEngine.activation_range_exceeded().
 * STICKY ERROR: every forward ends with an asynchronous copy of the flag word into pinned host memory (no synchronisation), and
 * every forward BEGINS by looking at that word: once a forward has raised the flag, the next rr_forward* call on the handle
 * returns RR_ERR_RANGE instead of computing — a caller that never polls cannot keep ranking with out-of-range activations for
 * more than the one batch whose logits it was about to read anyway.  rr_activation_range_flag(reset = 1) clears it (then build
 * the handle with compute_dtype = bf16 for this checkpoint).
 * The 9e8 limit applies to compute_dtype = fp16 handles only.  A bf16 handle (fp32's exponent range) raises the flag, and is
 * refused afterwards, only when a row is NOT FINITE (inf / NaN in the inputs or the weights); 3e4 is an ordinary value there.
 */
int rr_activation_range_flag(rr_handle h, int reset, int* flag_out, void* hip_stream);

/* rr_forward_packed: the same computation over PACKED rows (SURVEY.md "Variable length"): the caller
 * groups the pairs into n_segments segments of equal row length seg_len[i] <= padded_seq_len (the length the reference
 * would pad to, utils.py:157-165; every pair's non-pad tokens must fit its segment's length) and hands over
 *   input_ids / attention_mask / token_type_ids : DEVICE int64, sum_i seg_pairs[i] * seg_len[i] entries: segment after
 *       segment, inside a segment pair after pair, seg_len[i] entries per pair (token_type_ids may be NULL);
 *   image_cls / image_patches : DEVICE float32 PER PAIR, [n, vision_hidden] / [n, n_patches, vision_hidden] in the packed
 *       pair order, or both NULL (a segment mixes the candidates of several queries);
 *   logits_out (/ logits2_out for 2H_BCE) : DEVICE float32 [n] in the packed pair order; the caller scatters them back
 *       and runs rr_head on the [Bq, K] block.
 * Every GEMM, every LayerNorm statistics pass and the attention of a layer run ONCE over all rows of the call (attention as
 * one launch over the segments' workgroups when the padded call's grid selects the fixed-reference schedule); the embedding
 * gathers and the CLS heads run once per segment.  A pair's logit equals what rr_forward computes for it after
 * rr_set_padded_seq_len(padded_seq_len) at seq_len = seg_len[i], hence what the padded call computes: bit for bit for
 * text-only models (attention runs the schedule the padded call's grid would choose, whatever the segment's size); with
 * vision tokens up to fp32 summation order in the cross-encoder's attention and in the per-pair vision GEMMs (1.3e-4 at
 * 800 pairs, tests/test_gpu_parity_fullsize.py).  For rr_reserve count n_queries = n_pairs (image features are per pair).
 * No reference counterpart (it pads); Python: RerankEngine.forward_ids_packed.  RR_ERR_BAD_SHAPE: more than 64 segments,
 * an empty segment, a length above padded_seq_len or, with image features, below the mapping network's cross-attention
 * window (32). */
int rr_forward_packed(rr_handle h, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids,
                      const float* image_cls, const float* image_patches, int n_segments, const int32_t* seg_pairs,
                      const int32_t* seg_len, int padded_seq_len, float* logits_out, float* logits2_out, void* hip_stream);

/* rr_forward: one pass of the hot path over N = Bq*K (query,candidate) pairs.
 *   input_ids, attention_mask, token_type_ids : DEVICE int64 [N,S] row-major, query-major pair order
 *       (prepare_full_context_inputs, utils.py:129-167).  attention_mask masks keys in the text
 *       encoder; the cross-encoder mask is (input_id != 0) (rerank_model.py:385-392,562).
 *   image_cls     : DEVICE float32 [Bq, vision_hidden] or NULL (text_only) — CLIP last_hidden_state[:,0]
 *   image_patches : DEVICE float32 [Bq, n_patches, vision_hidden] or NULL — CLIP hidden_states[-2][:,1:]
 *       (per query; repeated over the K pairs inside, rerank_model.py:541-544)
 *   labels        : DEVICE float32 [N] or NULL (NULL: first candidate of each query is the positive,
 *       utils.py:239-243; must be NULL for RR_LOSS_NEGATIVE_SAMPLING, utils.py:233)
 *   logits_out    : DEVICE float32 [N]   (= [N,1] pointwise / [Bq,K] listwise; 2H_BCE: second head)
 *   logits2_out   : DEVICE float32 [N] or NULL (first head of 2H_BCE / classifier2 otherwise)
 *   loss_out      : DEVICE float32 [1] or NULL
 *   scores_out    : DEVICE float32 [N] or NULL — sigmoid(logit) (pointwise) / softmax over K (listwise)
 *   order_out     : DEVICE int32 [Bq,K] or NULL — per query, candidate indices sorted by logit,
 *       descending, ties in retrieval order (Reranker_base_executor.py:934-935)
 *   pair_begin/pair_end : this rank's contiguous slice [pair_begin, pair_end) of the N pairs
 *       (multi-GPU sharding; 0,N = everything).  Only that slice of logits_out is written; the head
 *       (loss/scores/order) is computed by rr_head after the caller has all-gathered logits.
 *       With the full range the head runs inside rr_forward.
 */
int rr_forward(rr_handle h, const int64_t* input_ids, const int64_t* attention_mask,
               const int64_t* token_type_ids, const float* image_cls, const float* image_patches,
               int Bq, int K, int S, const float* labels, int pair_begin, int pair_end,
               float* logits_out, float* logits2_out, float* loss_out, float* scores_out,
               int32_t* order_out, void* hip_stream);

/* rr_encode_image: the frozen CLIP vision tower the rerankers call once per query
 * (rerank_model.py:408-411,424-426 -> FLMRVisionModel.forward, modeling_flmr.py:1701-1757 -> CLIPVisionTransformer):
 * patch convolution (stride = kernel, no bias) as an im2col GEMM, [class | patches] + position embedding,
 * pre_layrnorm, vit_layers pre-LN blocks with quick-GELU MLPs.  Weights: the reference state_dict keys
 * "context_vision_encoder.vision_model.vision_model.*" (required when cfg.vit_layers > 0; post_layernorm is not read).
 *   pixel_values      : DEVICE float32 [B, 3, vit_image_size, vit_image_size]
 *   image_cls_out     : DEVICE float32 [B, vision_hidden]            = last_hidden_state[:, 0] (no post_layernorm)
 *   image_patches_out : DEVICE float32 [B, n_patches, vision_hidden] = hidden_states[-2][:, 1:]
 * The outputs are exactly the image_cls / image_patches arguments of rr_forward / rr_forward_joint. */
int rr_encode_image(rr_handle h, const float* pixel_values, int B, float* image_cls_out, float* image_patches_out,
                    void* hip_stream);

/* rr_forward_joint: RerankModel.forward (the "softmax"/2-head variant, rerank_model.py:171-331) from the joint
 * sequence the caller assembled as the reference does (:204-224): joint_ids = cat(query_ids repeated K times,
 * context_ids[:, 2 : 2 - query_len]) [N,S], same for the mask.  Inside: token types 0, query_mask with instruction
 * masking (id != 0 and (pos > first instruction_token_id position or pos < 2), :481-506; instruction_token_id < 0
 * = plain id != 0), cross-encoder token order [query | image | context] (:257-274), and the reference's
 * `loss_fn(logits, logits)` (:328: labels are ignored, the loss uses the logits as targets).  Image features are
 * mandatory (NotImplementedError for text_only, :184-185).  Attention fusion: rr_forward_joint_fusion. */
int rr_forward_joint(rr_handle h, const int64_t* joint_input_ids, const int64_t* joint_attention_mask,
                     const float* image_cls, const float* image_patches, int Bq, int K, int S, int query_len,
                     int64_t instruction_token_id, int pair_begin, int pair_end, float* logits_out,
                     float* logits2_out, float* loss_out, float* scores_out, int32_t* order_out, void* hip_stream);

/* rr_forward_joint_fusion: rr_forward_joint with the PreFLMR attention fusion of RerankModel.forward
 * (`preflmr_scores`, `fusion_multiplier`; rerank_model.py:276-319, attention_fusion.py:84-102; the executor passes the
 * retriever's `scores_raw`, Reranker_base_executor.py:888-891).
 *   preflmr_scores : DEVICE float32 [N, S, query_len + prefix_len + n_patches] — context token x query/image token.
 * Rows 2 .. 2 + S - query_len of every pair (the context tokens that are in the joint sequence) give an additive
 * attention bias over the cross-encoder tokens [query | image | context] in every cross-encoder layer: query rows get
 * softmax-over-context-tokens, context rows softmax-over-query-tokens, the self blocks 0, all times fusion_multiplier. */
int rr_forward_joint_fusion(rr_handle h, const int64_t* joint_input_ids, const int64_t* joint_attention_mask,
                            const float* image_cls, const float* image_patches, const float* preflmr_scores,
                            float fusion_multiplier, int Bq, int K, int S, int query_len, int64_t instruction_token_id,
                            int pair_begin, int pair_end, float* logits_out, float* logits2_out, float* loss_out,
                            float* scores_out, int32_t* order_out, void* hip_stream);

/* rr_forward_interaction: the Interaction rerankers, fed by the frozen retriever's late-interaction outputs
 * (Reranker_base_executor.py:877-885 builds the call).  All pointers DEVICE float32:
 *   query_li [Bq, Lq, li_dim], context_li [N, Lc, li_dim], query_mask [Bq, Lq] and context_mask [N, Lc] (0/1).
 * Outputs, labels, pair slice and head semantics are those of rr_forward.  (No attention fusion here: the reference's
 * interaction executor path never passes it and MORES raises NotImplementedError for it, mores_model.py:72-73.) */
int rr_forward_interaction(rr_handle h, const float* query_li, const float* context_li, const float* query_mask,
                           const float* context_mask, int Bq, int K, int Lq, int Lc, const float* labels,
                           int pair_begin, int pair_end, float* logits_out, float* logits2_out, float* loss_out,
                           float* scores_out, int32_t* order_out, void* hip_stream);

/* rr_forward_interaction_fusion: the same with the PreFLMR attention fusion of InteractionRerankModel.forward
 * (interaction_rerank_model.py:131-142; NORMAL interaction type only, MORES raises NotImplementedError,
 * mores_model.py:72-73): preflmr_scores DEVICE float32 [N, Lc, Lq] (context token x query token) becomes the additive
 * bias [[0, softmax(scores^T)], [softmax(scores), 0]] * fusion_multiplier over the tokens [query | context]. */
int rr_forward_interaction_fusion(rr_handle h, const float* query_li, const float* context_li, const float* query_mask,
                                  const float* context_mask, const float* preflmr_scores, float fusion_multiplier, int Bq,
                                  int K, int Lq, int Lc, const float* labels, int pair_begin, int pair_end,
                                  float* logits_out, float* logits2_out, float* loss_out, float* scores_out,
                                  int32_t* order_out, void* hip_stream);

/* rr_head: scoring head + loss + top-K order over complete logits [Bq*K] (after the RCCL
 * all-gather of per-rank slices).  Same semantics as the tail of rr_forward. */
int rr_head(rr_handle h, const float* logits, const float* logits2, const float* labels, int Bq, int K,
            float* loss_out, float* scores_out, int32_t* order_out, void* hip_stream);

/* ---- Host-side pair-input assembly (no GPU involved): WordPiece tokenisation of the (query, candidate) texts into the
 * int64 [N, S] tensors rr_forward takes.  Replaces prepare_full_context_inputs (src/models/rerank/utils.py:129-167) and
 * the BertTokenizer encode / decode / batch_encode_plus calls under it (transformers 4.38.2 slow tokenizer semantics).
 * All pointers HOST.  A handle is immutable after creation and may be shared by threads. */
typedef struct rr_tokenizer* rr_tokenizer_handle;
/* vocab_tokens[i] = UTF-8 token with id i (the lines of vocab.txt); must contain [UNK] [CLS] [SEP] [PAD]. */
int rr_tok_create(const char* const* vocab_tokens, int vocab_size, int do_lower_case, rr_tokenizer_handle* out);
int rr_tok_destroy(rr_tokenizer_handle h);
/* encode(text, add_special_tokens=False, max_length=max_tokens, truncation=True); max_tokens < 0 = no limit.
 * Returns the number of ids written, or <0 (RR_ERR_BAD_SHAPE: capacity too small). */
int rr_tok_encode(rr_tokenizer_handle h, const char* text, int max_tokens, int32_t* ids_out, int capacity);
/* decode(ids) with clean_up_tokenization_spaces=True; returns the byte length (NUL-terminated), or <0. */
int rr_tok_decode(rr_tokenizer_handle h, const int32_t* ids, int n, char* out, int capacity);
/* The whole of prepare_full_context_inputs: queries[n_queries], contexts[n_queries * docs_per_query] (query-major);
 * every text is truncated by an encode -> decode -> encode round trip (max_query_length / max_context_length tokens),
 * pairs are encoded [CLS] q [SEP] c [SEP] with LONGEST_FIRST truncation to max_length and right padding (pad id,
 * mask 0, type 0).  Outputs int64 [N, max_length], e.g. pinned buffers; n_threads <= 0 = hardware concurrency. */
int rr_tok_prepare_pairs(rr_tokenizer_handle h, const char* const* queries, int n_queries, const char* const* contexts,
                         int docs_per_query, int max_query_length, int max_context_length, int max_length, int n_threads,
                         int64_t* input_ids, int64_t* attention_mask, int64_t* token_type_ids);

/* Profiling: when on, rr_forward brackets every kernel launch with HIP events on the work
 * stream; rr_get_profile synchronises, accumulates and returns the per-class totals. */
int rr_set_profiling(rr_handle h, int on);
int rr_get_profile(rr_handle h, rr_profile* out, int reset);

/* Diagnostic and test-support entry points (debug taps, stand-alone operators rr_op_*, process-wide tuning switches, in-kernel
 * timelines) are exported by the same library but are NOT part of the product ABI: include/rerank_mi355_diag.h. */

#ifdef __cplusplus
}
#endif
#endif /* RERANK_MI355_H */
