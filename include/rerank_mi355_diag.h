/*
 * rerank_mi355_diag.h — diagnostic and test-support entry points of librerank_mi355.so
 *
 * NOT the product ABI.  include/rerank_mi355.h is what a binding of the reference's reranker binds; the
 * functions below exist for the parity tests (stand-alone operators, debug taps), the A/B tools under tools/ (process-wide
 * switches that select between equivalent kernels or move time) and the in-kernel timelines.  They are process-wide, not
 * thread-safe against running forwards, carry no compatibility promise and are never needed on the product path: what
 * changes the arithmetic of a forward is a per-handle option (rr_set_option in rerank_mi355.h).
 * The arithmetic they expose follows stock HF BERT behind /root/reference/src/models/flmr/models/flmr/modeling_flmr.py:1622
 * (text encoder) and /root/reference/src/models/rerank/attention_fusion.py:133-144 (cross-encoder layers).
 */
#ifndef RERANK_MI355_DIAG_H
#define RERANK_MI355_DIAG_H

#include "rerank_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug taps: copy an internal activation of the LAST rr_forward to HOST memory as float32.
 * names: "text_hidden" [n,S,H], "late_interaction" [n,T,D], "ce_hidden" [n,T,Hc]. Returns element
 * count written, or <0.  Synchronises the stream.  Test-only. */
int64_t rr_debug_read(rr_handle h, const char* name, float* host_out, int64_t max_elems);

int rr_set_debug(rr_handle h, int on);   /* keep a copy of the text-encoder output for rr_debug_read */

/* Stand-alone operator entry points (unit parity tests call the kernels through these).
 * All pointers DEVICE.  bf16 tensors are uint16_t bit patterns.  Kd % 64 == 0, N % 4 == 0. */
int rr_op_gemm_bf16(const uint16_t* A /*[M,Kd]*/, const uint16_t* W /*[N,Kd]*/, const float* bias /*[N]|NULL*/,
                    int M, int N, int Kd,
                    int epilogue /*0: +bias -> bf16; 1: +bias, erf-GELU -> bf16; 2: +bias -> f32; 3: +bias, tanh -> bf16;
                                   5: +bias, quick-GELU -> bf16*/,
                    void* out, void* hip_stream);
/* out f32 [M,N] = A W^T + bias + resid */
int rr_op_gemm_resid_f32(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid /*[M,N]*/,
                         int M, int N, int Kd, float* out, void* hip_stream);
/* The LayerNorm-statistics dataflow the layers use between blocks: rr_op_layernorm_stats writes the 16-bit normalised
 * rows (and optionally the fp32 ones) plus stats[row] = (mean, rstd); rr_op_gemm_ln_resid_f32 then forms its residual
 * as (x - mean) * rstd * gamma + beta from the LayerNorm's INPUT x:  out = A W^T + bias + LN(x). */
int rr_op_layernorm_stats(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols,
                          float* out_f32 /*|NULL*/, uint16_t* out_bf16, float* stats /*[rows,2]*/, void* hip_stream);
int rr_op_gemm_ln_resid_f32(const uint16_t* A, const uint16_t* W, const float* bias, const float* x /*[M,N]*/,
                            const float* stats /*[M,2]*/, const float* gamma /*[N]*/, const float* beta /*[N]*/, int M,
                            int N, int Kd, float* out, void* hip_stream);
/* softmax(q k^T + key_bias) v per head (head dim 64; q is expected pre-scaled by log2(e)/sqrt(64): the kernel takes
 * base-2 exponentials of q k^T as it is; key_bias is additive in that domain, 0 / -1e30).
 * q row (b,t): q + ((b / q_batch_div) * Tq + t) * q_stride + head*64 ; k,v row (b,t): (b*Tk + t) * kv_stride + head*64;
 * key_bias f32 [B,Tk] additive (0 = attend, -1e30 = masked) or NULL.  q_stride, kv_stride and out_stride are multiples of 8
 * elements (16-byte row chunks). */
int rr_op_attention_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int q_stride, int kv_stride,
                         const float* key_bias, int B, int heads, int Tq, int Tk, int q_batch_div, uint16_t* out,
                         int out_stride, void* hip_stream);
/* Tuning hooks (tools/bench_gemm.py). rr_set_gemm_variant forces a kernel / tile configuration: 0..3 = gemm_kernel_s
 * {128x128x2st, 128x128x4st, 256x256x2st, 256x128x3st} with the direct epilogue, 10 = 256x256x2st with the LDS-staged
 * epilogue, 11 / 12 = gemm_kernel_h (half-tile ring) with the direct / LDS-staged epilogue, 13 = its diagnostic timeline
 * build, 14 = gemm_kernel_hp (persistent ring); -1 = shape heuristic (default).  rr_set_gemm_stamps: DEVICE buffer of
 * 8 uint64 per workgroup that receives s_memtime stamps (entry, first tile ready, main loop done, end), or NULL.
 * Both are process-wide and diagnostic. */
/* e4m3 (OCP fp8) GEMM on the block-scaled matrix core (v_mfma_scale_f32_16x16x128_f8f6f4, block scales 2^0):
 * out = epi(scale * A8[M,K] . W8[N,K]^T + bias), A8/W8 row-major e4m3 bytes, scale = the product of the two per-tensor
 * dequantisation scales, epilogue 0 = bf16 out, 1 = bf16(erf-GELU), 2 = f32 out.  K % 128 == 0, N % 4 == 0.
 * Per-tensor-scale form (BASELINE configs[4]); the forward uses rr_op_gemm_fp8_rc's scaling. */
int rr_op_gemm_fp8(const uint8_t* A8, const uint8_t* W8, const float* bias, float scale, int M, int N, int K, int epilogue,
                   void* out, void* hip_stream);
/* The form the model forward uses: out = epi(row_scale[m] * col_scale[n] * (A8 . W8^T) + bias), activations quantised per row
 * (rr_op_layernorm_q8), weights per output channel; either scale vector may be NULL (= 1).  Large problems run the
 * persistent ring on v_mfma_scale_f32_32x32x64_f8f6f4, small ones the two-stage kernel. */
int rr_op_gemm_fp8_rc(const uint8_t* A8, const uint8_t* W8, const float* bias, const float* row_scale, const float* col_scale,
                      int M, int N, int K, int epilogue, void* out, void* hip_stream);
/* The two GEMMs of the fp8 configuration's FFN on the persistent e4m3 ring (shapes with at least 512 tiles of 256 x 256,
 * K % 128 == 0, N % 16 == 0; RR_ERR_BAD_SHAPE otherwise):
 *   rr_op_gemm_fp8_gelu_e4m3: out8[M,N] = e4m3(clamp(out_mul * gelu(row_scale[m] * col_scale[n] * (A8 . W8^T) + bias), +-448)):
 *     the GELU output under ONE static scale (the forward uses 8), the A operand of
 *   rr_op_gemm_fp8_resid: out[M,N] (f32) = scale * col_scale[n] * (A8 . W8^T) + bias + r, r = resid[M,N] when stats == NULL,
 *     else gamma * (resid - mean_m) * rstd_m + beta with stats[m] = (mean, rstd) — the fp32-stream residual epilogue of
 *     rr_op_gemm_ln_resid_f32 behind an e4m3 GEMM.  Reference seam: BertIntermediate / BertOutput of stock HF BERT. */
int rr_op_gemm_fp8_gelu_e4m3(const uint8_t* A8, const uint8_t* W8, const float* bias, const float* row_scale, const float* col_scale,
                             float out_mul, int M, int N, int K, uint8_t* out8, void* hip_stream);
int rr_op_gemm_fp8_resid(const uint8_t* A8, const uint8_t* W8, const float* bias, float scale, const float* col_scale,
                         const float* resid, const float* stats, const float* gamma, const float* beta, int M, int N, int K, float* out,
                         void* hip_stream);
/* Per-tensor e4m3 quantisation for rr_op_gemm_fp8: out[i] = e4m3(clamp(x[i] / scale, +-448)), round to nearest even;
 * x holds n (a multiple of 8) f32 values (x_is_f32 != 0) or bf16 values.  rr_op_amax: *out_dev (device float) = max |x|
 * (exact and order-independent), from which the caller derives scale = amax / 448. */
int rr_op_quantize_fp8(const void* x, int x_is_f32, float scale, uint8_t* out, size_t n, void* hip_stream);
int rr_op_amax(const void* x, int x_is_f32, size_t n, float* out_dev, void* hip_stream);
/* LayerNorm folded into the consumer GEMM (north_star "fused LayerNorm+QKV"), the two halves stand-alone:
 *   rr_op_gemm_resid_lnprep: out_f32 = A W^T + bias + resid (as rr_op_gemm_resid_f32) and, from the same epilogue, x16_out =
 *     the 16-bit copy of those rows plus per-row LayerNorm statistics stats_out[row] = (mean, rstd) of out_f32's rows
 *     (merged from per-128-column partials in part_scratch [M, ceil(N/128), 2]).  N % 8 == 0.
 *   rr_op_gemm_lnfold: out = epi(rstd_m * (A_raw W_folded^T - mean_m * csum) + dvec), epilogue 0 = 16-bit, 1 = 16-bit erf-GELU,
 *     2 = f32; with W_folded = 16bit(W * gamma), csum_n = sum_k W_folded[n,k], dvec = W beta + b this is
 *     epi(LayerNorm(x) W^T + b) for the raw rows x whose 16-bit copy is A_raw. */
int rr_op_gemm_resid_lnprep(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid, int M, int N, int Kd,
                            float eps, float* out_f32, uint16_t* x16_out, float* stats_out, float* part_scratch, void* hip_stream);
/* The same residual epilogue on the SPLIT residual stream: a pre-LayerNorm row x travels as hi = its 16-bit
 * operand rounding (the consumer GEMM's A rows) + lo = fp16(x - hi) instead of a separate fp32 copy.  Residual rows come in as
 * (hi_in, lo_in) [M,N] — normalised on the fly with (ln_stats [M,2], ln_gamma, ln_beta) when ln_stats != NULL — and the output
 * rows x = A W^T + bias + residual leave as (x16_out, lo_out) plus their statistics; hi_in == x16_out and lo_in == lo_out
 * (in place) is allowed.  With the process-wide "resid_lo8" in effect (rr_set_tuning; default: on for the fp16 operand type) lo_in /
 * lo_out are e5m2 BYTES of (x - hi) * 16 in the epilogues' private layout — rows r and r + 16 of an aligned 32-row group
 * interleaved in units of 8 columns, ceil(M / 32) * 32 rows of N bytes (csrc/rr_common.h lo8_pair_offset; tests/test_gpu_ops.py
 * _lo8_to_device_layout).  Only for shapes the persistent ring kernel runs (>= 128 tiles of 256 x 256 — rr_set_tuning "gemm_ring_min_tiles" —, N % 8 == 0), otherwise
 * RR_ERR_UNSUPPORTED. */
int rr_op_gemm_resid_split(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* hi_in, const uint16_t* lo_in,
                           const float* ln_stats, const float* ln_gamma, const float* ln_beta, int M, int N, int Kd, float eps,
                           uint16_t* x16_out, uint16_t* lo_out, float* stats_out, float* part_scratch, void* hip_stream);
/* Test support: the fp32 residual rows the split-stream epilogue forms from a (hi, lo) pair — hi + lo, LayerNorm-recomputed when
 * `stats` (mean, rstd per row) / gamma / beta are given — with the epilogue's own expression, so that rr_op_gemm_resid_lnprep fed
 * these rows is a bit-exact expectation for rr_op_gemm_resid_split (tests/test_gpu_ops.py).  hi in the operand type of
 * rr_set_op_dtype, lo fp16; cols even. */
int rr_op_split_residual_value(const uint16_t* hi, const uint16_t* lo, const float* stats, const float* gamma, const float* beta,
                               int rows, int cols, float* out, void* hip_stream);
int rr_op_gemm_lnfold(const uint16_t* A_raw, const uint16_t* W_folded, const float* dvec, const float* csum, const float* stats,
                      int M, int N, int Kd, int epilogue, void* out, void* hip_stream);
/* LayerNorm whose output is an fp8 GEMM operand: out8[row] = e4m3(LN(x[row]) / row_scale[row]), row_scale = row amax / 448,
 * stats (may be NULL) = (mean, rstd).  rr_util_quantize_rows_e4m3 is the HOST routine the weight packer uses (per output
 * channel = per row of W [rows, cols]): usable without a GPU. */
int rr_op_layernorm_q8(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols, uint8_t* out8,
                       float* row_scale, float* stats, void* hip_stream);
int rr_util_quantize_rows_e4m3(const float* w_host, int rows, int cols, uint8_t* out_host, float* scales_host);
int rr_set_gemm_variant(int variant);
/* Process-wide DIAGNOSTIC switches (A/B tools, tests): the default every handle option of the same name follows until
 * rr_set_option pins it ("ln_lite", "ln_fold", "resid_split", "resid_lo8", "ce_cls_only", "fp8_ffn_down", "attn_fixed_ref": see rr_set_option),
 * plus switches that select between bit-identical kernels or only move time: "resid_fast" (default 1: plain fp32 residual GEMMs
 * on the split forms' epilogue), "resid_touch" (0: L2 touch of the next residual pass), "gemm_desync" (0: start skew of the XCDs,
 * percent of a tile period), "persistent_gemm" (1), "gemm_ring_min_tiles" (128: smallest problem, in 256 x 256 tiles, on the
 * persistent ring), "m_alternate" (1: consecutive large launches of the layer chain walk the rows in opposite directions, so that
 * a consumer starts on the rows its producer wrote last; results bit-identical either way),
 * "attn_prio" (1).  "resid_lo8": -1 (the built-in default) = by operand type (1 for fp16, 0 for bf16), 0 / 1 = for every handle that has not
 * pinned it; the environment variable RR_RESID_LO8 = 0 | 1, read once at load, replaces the built-in -1 (lets an unmodified test
 * run take either form).  Not thread-safe against running forwards; never needed on the product path. */
int rr_set_tuning(const char* key, int value);
int rr_set_op_dtype(int dt);          /* operand dtype (0 bf16 / 1 fp16) of the stand-alone rr_op_* entry points */
int rr_set_gemm_stamps(void* device_buf);
int rr_set_attn_stamps(void* device_buf);   /* diagnostic timeline of the attention kernel: 4 x 8 uint64 per workgroup, or NULL */
int rr_set_attn_redo_stats(void* device_buf);   /* diagnostic: DEVICE 2 x uint64 — the redo launches of the fixed-reference attention add (workgroups flagged for the online recompute, workgroups looked at) — or NULL */
int rr_set_gemm_stagger(int unit);   /* diagnostic codes of the 16-bit GEMM kernels, 0 = none: 1..49 start skew of the first dispatch wave in
                                       s_sleep(127) units; 50..55 tile-order groups of 2..64 row panels, 56 row-major, 57 = round 4's rule (groups of 8 also for N <= 1024); 59 = the persistent ring
                                       WITHOUT its serpentine K walk (every second block of 1 024 output columns accumulates its K-tiles from the
                                       last to the first: gemm_bf16.hip k_walk_reversed; with 59 the ring no longer agrees to the bit with the
                                       simple kernel, which keeps the rule); 61..64 timeline builds only */
int rr_op_layernorm(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols,
                    float* out_f32, uint16_t* out_bf16, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RERANK_MI355_DIAG_H */
