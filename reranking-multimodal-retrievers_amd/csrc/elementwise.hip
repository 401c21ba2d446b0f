// HBM-bound row kernels of the rerank path (gfx950): embeddings + LayerNorm, LayerNorm, the
// late-interaction bottleneck's mask / L2-normalise / concat, cross-encoder input embeddings,
// key-padding bias, small gathers/casts, CLS classifier heads.  One wave (64 lanes) owns one row,
// 16-byte vector loads/stores, wave-shuffle reductions; fp32 statistics throughout.
//
// Reference anchors (/root/reference/): BertEmbeddings + LayerNorm as called through
// modeling_flmr.py:1622 and attention_fusion.py:126-132; mask / normalise rerank_model.py:385-392,
// 471-478; classifier heads utils.py:101-108.
#include "rr_common.h"

namespace {

constexpr int MAX_V4 = 8;   // row length <= 64 lanes * 8 float4 = 2048 columns

struct RowStats { float mean, rstd; };

// v[i] holds float4 index lane + 64 i of the row (zeros beyond n4).
__device__ __forceinline__ RowStats row_stats(const float4 (&v)[MAX_V4], int n4, int lane, int cols, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i)
    if (lane + 64 * i < n4) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i)
    if (lane + 64 * i < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  const float var = wave_sum(q) / (float)cols;
  return RowStats{mean, 1.0f / sqrtf(var + eps)};
}

__device__ __forceinline__ void ln_store(const float4 (&v)[MAX_V4], RowStats st, const float* gamma,
                                         const float* beta, int n4, int lane, float* o32, bf16_t* o16, int dt) {
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      const float4 g = ((const float4*)gamma)[c4], b = ((const float4*)beta)[c4];
      float4 y;
      y.x = (v[i].x - st.mean) * st.rstd * g.x + b.x;
      y.y = (v[i].y - st.mean) * st.rstd * g.y + b.y;
      y.z = (v[i].z - st.mean) * st.rstd * g.z + b.z;
      y.w = (v[i].w - st.mean) * st.rstd * g.w + b.w;
      if (o32) ((float4*)o32)[c4] = y;
      if (o16) ((uint2*)o16)[c4] = make_uint2(pack2rt(y.x, y.y, dt), pack2rt(y.z, y.w, dt));
    }
  }
}

// ---- LayerNorm over fp32 rows -> fp32 (residual stream) + bf16 (next GEMM operand)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int rows,
                                                        int cols, float* __restrict__ o32, bf16_t* __restrict__ o16, int dt,
                                                        float2* __restrict__ stats_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = cols >> 2;
  float4 v[MAX_V4];
  const float4* xr = (const float4*)(x + (size_t)row * cols);
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) v[i] = (lane + 64 * i < n4) ? xr[lane + 64 * i] : make_float4(0, 0, 0, 0);
  const RowStats st = row_stats(v, n4, lane, cols, eps);
  if (stats_out && lane == 0) stats_out[row] = make_float2(st.mean, st.rstd);   // for the GEMM epilogue's ln_apply
  ln_store(v, st, gamma, beta, n4, lane, o32 ? o32 + (size_t)row * cols : nullptr,
           o16 ? o16 + (size_t)row * cols : nullptr, dt);
}

// ---- folded LayerNorm: merge the per-128-column-group (mean, M2) partials a residual GEMM's epilogue left behind into
// (mean, rstd) per row.  Equal-weight groups except the last: Chan et al. pairwise update, evaluated in group order
// (deterministic).  One thread per row; rows x nparts x 8 bytes in, rows x 8 bytes out.
// Range guard: these rows travel as 16-bit operand rows (fp16 in the headline mode: largest finite value
// 65 504).  max |x| <= sqrt(sum x^2) = sqrt(M2 + n mean^2), so a row whose sum of squares stays below RANGE_SS cannot hold
// an element above 3e4; a row that reaches it (or is not finite) raises *range_flag — the statistics are taken from the
// fp32 values before the 16-bit rounding, so the test still works when the rounding has already overflowed.
// The limit is the caller's: RR_RANGE_SS_FP16 for fp16 operand rows; +inf for bf16 rows, whose exponent range is
// fp32's — there only a NON-FINITE row raises the flag (`!(x < inf)` holds for inf and NaN alone).
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ part, int nparts, int cols, float eps,
                                                          int rows, float2* __restrict__ stats, int* __restrict__ range_flag,
                                                          float range_ss) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* p = part + (size_t)row * nparts;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int g = 0; g < nparts; ++g) {
    const float2 v = p[g];
    const float nb = (float)min(128, cols - g * 128), nt = n + nb;
    const float delta = v.x - mean;
    mean += delta * (nb / nt);
    m2 += v.y + delta * delta * (n * nb / nt);
    n = nt;
  }
  stats[row] = make_float2(mean, 1.0f / sqrtf(m2 / n + eps));
  if (range_flag && !(m2 + n * mean * mean < range_ss)) atomicOr(range_flag, 1);      // (NaN fails the comparison too)
}

// ---- LayerNorm whose output is the e4m3 operand of an fp8 GEMM (BASELINE configs[4]): y = LN(x) in fp32, one scale per
// ROW (row amax / 448: the whole e4m3 range is used whatever the row's magnitude; no calibration state), y / scale rounded
// to nearest even by v_cvt_pk_fp8_f32 (OCP e4m3fn on gfx950).  Also leaves (mean, rstd) for the residual epilogue's ln_apply.
// 5 bytes per element instead of 6.  A zero row gets scale 1 (all-zero codes).
__global__ __launch_bounds__(256) void layernorm_q8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int rows, int cols,
                                                           uint8_t* __restrict__ o8, float* __restrict__ row_scale,
                                                           float2* __restrict__ stats_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = cols >> 2;
  float4 v[MAX_V4];
  const float4* xr = (const float4*)(x + (size_t)row * cols);
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) v[i] = (lane + 64 * i < n4) ? xr[lane + 64 * i] : make_float4(0, 0, 0, 0);
  const RowStats st = row_stats(v, n4, lane, cols, eps);
  if (stats_out && lane == 0) stats_out[row] = make_float2(st.mean, st.rstd);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      const float4 g = ((const float4*)gamma)[c4], b = ((const float4*)beta)[c4];
      v[i].x = (v[i].x - st.mean) * st.rstd * g.x + b.x;
      v[i].y = (v[i].y - st.mean) * st.rstd * g.y + b.y;
      v[i].z = (v[i].z - st.mean) * st.rstd * g.z + b.z;
      v[i].w = (v[i].w - st.mean) * st.rstd * g.w + b.w;
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    }
  }
  amax = wave_max(amax);
  const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / scale;
  if (lane == 0) row_scale[row] = scale;
  // A lane's four codes of column group c4 are one dword; stored directly that is 4 bytes per lane and instruction.  The
  // row is gathered in a wave-private LDS line instead and leaves as 16 bytes per lane (same-wave LDS accesses execute in
  // issue order: no barrier).
  __shared__ __attribute__((aligned(16))) uint32_t line[4][64 * MAX_V4];
  uint32_t* const ln_ = line[threadIdx.x >> 6];
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      const float a = __builtin_amdgcn_fmed3f(v[i].x * inv, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(v[i].y * inv, -448.f, 448.f);
      const float c = __builtin_amdgcn_fmed3f(v[i].z * inv, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(v[i].w * inv, -448.f, 448.f);
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
      ln_[c4] = (uint32_t)w;
    }
  }
  uint8_t* const orow = o8 + (size_t)row * cols;
  const bool wide = ((cols & 15) == 0) && ((((size_t)orow) & 15) == 0);
#pragma unroll
  for (int j = 0; j < (MAX_V4 + 3) / 4; ++j) {
    const int d0 = (lane + 64 * j) * 4;                 // first dword of this lane's 16-byte chunk
    if (d0 >= n4) continue;
    if (wide) {
      *(uint4*)(orow + (size_t)d0 * 4) = *(const uint4*)(ln_ + d0);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (d0 + k < n4) ((uint32_t*)orow)[d0 + k] = ln_[d0 + k];
    }
  }
}

// ---- BertEmbeddings from ids: word[id] + type[tt] + pos[s] -> LN
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tts,
                                                       const float* __restrict__ word, const float* __restrict__ pos,
                                                       const float* __restrict__ type, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int rows, int S,
                                                       int cols, int vocab, int type_vocab,
                                                       float* __restrict__ o32, bf16_t* __restrict__ o16, int dt) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = cols >> 2;
  int64_t id = ids[row], tt = tts ? tts[row] : 0;
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);           // host validates; clamp keeps loads in bounds
  tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
  const float4* wr = (const float4*)(word + (size_t)id * cols);
  const float4* tr = (const float4*)(type + (size_t)tt * cols);
  const float4* pr = (const float4*)(pos + (size_t)(row % S) * cols);
  float4 v[MAX_V4];
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      const float4 a = wr[c4], b = tr[c4], c = pr[c4];
      // same association as HF: (inputs_embeds + token_type_embeddings) + position_embeddings
      v[i] = make_float4((a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z, (a.w + b.w) + c.w);
    } else {
      v[i] = make_float4(0, 0, 0, 0);
    }
  }
  const RowStats st = row_stats(v, n4, lane, cols, eps);
  ln_store(v, st, gamma, beta, n4, lane, o32 + (size_t)row * cols, o16 + (size_t)row * cols, dt);
}

// ---- cross-encoder embeddings from inputs_embeds: x + type[0] + pos[p(t)] -> LN  (row = pair*T + t).  p(t) = t, except in a
// length-bucketed forward (rr_set_padded_seq_len): the pair's text occupies t < s_text and the tokens behind it (vision
// prefix + patches) keep the positions they have behind the PADDED text, vis_pos0 + (t - s_text).
__global__ __launch_bounds__(256) void ce_embed_ln_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                          const float* __restrict__ type0,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int rows, int T,
                                                          int cols, float* __restrict__ o32, bf16_t* __restrict__ o16,
                                                          int dt, int s_text, int vis_pos0, int cls32) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = cols >> 2;
  const float4* xr = (const float4*)(x + (size_t)row * cols);
  const int t = row % T, pt = t < s_text ? t : vis_pos0 + (t - s_text);
  const float4* pr = (const float4*)(pos + (size_t)pt * cols);
  const float4* tr = (const float4*)type0;
  float4 v[MAX_V4];
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      const float4 a = xr[c4], b = tr[c4], c = pr[c4];
      v[i] = make_float4((a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z, (a.w + b.w) + c.w);
    } else {
      v[i] = make_float4(0, 0, 0, 0);
    }
  }
  const RowStats st = row_stats(v, n4, lane, cols, eps);
  // cls32: the fp32 rows are wanted for the CLS row of each pair only (the cross-encoder's last layer on the CLS rows)
  ln_store(v, st, gamma, beta, n4, lane, (cls32 && t != 0) ? nullptr : o32 + (size_t)row * cols, o16 + (size_t)row * cols, dt);
}

// ---- late-interaction rows: (x * mask) -> L2 normalise (F.normalize eps 1e-12) -> bf16, scattered
// into the concatenated [pairs, T, D] buffer.  src row r = sb * rows_per_batch + j; it is written to
// every destination pair p with (p + pair_off) / bdiv == sb + src_batch_off... expressed from the
// destination side: one wave per destination row.
__global__ __launch_bounds__(256) void li_normalize_kernel(const float* __restrict__ src, const int64_t* __restrict__ ids,
                                                           int ids_stride, int n_pairs, int rows_per_batch, int D,
                                                           int T, int t_off, int pair_off, int bdiv,
                                                           int src_batch_off, bf16_t* __restrict__ dst, int dt,
                                                           int normalize, const float* __restrict__ maskf, int split,
                                                           int shift) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_pairs * rows_per_batch) return;
  const int p = r / rows_per_batch, j = r - p * rows_per_batch;
  const int sb = (p + pair_off) / bdiv - src_batch_off;
  const float* s = src + ((size_t)sb * rows_per_batch + j) * D;
  const float m = maskf ? maskf[(size_t)p * rows_per_batch + j]
                        : (ids ? (ids[(size_t)p * ids_stride + j] != 0 ? 1.0f : 0.0f) : 1.0f);
  const int n4 = D >> 2;
  float4 v[MAX_V4];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      float4 a = ((const float4*)s)[c4];
      a.x *= m; a.y *= m; a.z *= m; a.w *= m;
      v[i] = a;
      q += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
  }
  const float nrm = normalize ? fmaxf(sqrtf(wave_sum(q)), 1e-12f) : 1.0f;   // 0: plain convert/concat
  const int tj = t_off + j + (j >= split ? shift : 0);    // RerankModel reorders [query | image | context] (:257-264)
  bf16_t* d = dst + ((size_t)p * T + tj) * D;
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4)
      ((uint2*)d)[c4] = make_uint2(pack2rt(v[i].x / nrm, v[i].y / nrm, dt), pack2rt(v[i].z / nrm, v[i].w / nrm, dt));
  }
}

// ---- key-padding bias rows: text encoder from attention_mask, cross encoder from (id != 0) + ones
__global__ void key_bias_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ am, int n, int S, int T,
                                float* __restrict__ text_bias, float* __restrict__ ce_bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * T) return;
  const int p = i / T, t = i - p * T;
  if (t < S) {
    text_bias[(size_t)p * S + t] = am[(size_t)p * S + t] != 0 ? 0.f : -1e30f;
    ce_bias[i] = ids[(size_t)p * S + t] != 0 ? 0.f : -1e30f;
  } else {
    ce_bias[i] = 0.f;
  }
}

// ---- RerankModel (ids signature): query_mask with instruction masking (rerank_model.py:481-506) and the
// [query | image | context] reorder of the cross-encoder mask (:267-274).  One wave per pair.
//   valid(s) = id != 0 && (s > sep || s < 2), sep = first position of the instruction token (clamped to >= 1;
//   instruction_token < 0 switches the rule off).
__global__ __launch_bounds__(256) void joint_masks_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ am,
                                                          int n, int S, int P, int q_len, long long instruction_token,
                                                          float* __restrict__ text_bias, float* __restrict__ li_mask,
                                                          float* __restrict__ ce_bias) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int64_t* row = ids + (size_t)p * S;
  int sep = 0x7fffffff;
  if (instruction_token >= 0) {
    for (int s = lane; s < S; s += 64)
      if (row[s] == instruction_token) { sep = s; break; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sep = min(sep, __shfl_xor(sep, o, 64));
    if (sep == 0x7fffffff) sep = 0;      // torch.argmax of an all-zero row is 0 ...
    if (sep < 1) sep = 1;                // ... and positions < 1 are set to 1 (:491-493)
  }
  const int T = S + P;
  for (int s = lane; s < S; s += 64) {
    const bool valid = row[s] != 0 && (instruction_token < 0 || s > sep || s < 2);
    text_bias[(size_t)p * S + s] = am[(size_t)p * S + s] != 0 ? 0.f : -1e30f;
    li_mask[(size_t)p * S + s] = valid ? 1.f : 0.f;
    const int t = s < q_len ? s : s + P;
    ce_bias[(size_t)p * T + t] = valid ? 0.f : -1e30f;
  }
  for (int j = lane; j < P; j += 64) ce_bias[(size_t)p * T + q_len + j] = 0.f;
}

// ---- CLIP ViT front end (modeling_clip CLIPVisionEmbeddings as called through modeling_flmr.py:1701-1757):
// im2col for the stride = kernel patch convolution: row (b, patch), column (c, ky, kx) — the order of the conv weight
// [hidden, 3, ps, ps] flattened — zero-padded to Kp columns, written in the MFMA operand type.
__global__ void vit_im2col_kernel(const float* __restrict__ px, bf16_t* __restrict__ out, int B, int IS, int ps, int Kp,
                                  int dt) {
  const int g = IS / ps, np = g * g, Kd = 3 * ps * ps;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * np * Kp) return;
  const int col = (int)(i % Kp);
  const size_t rp = i / Kp;
  const int p = (int)(rp % np), b = (int)(rp / np);
  float v = 0.f;
  if (col < Kd) {
    const int c = col / (ps * ps), r = col - c * ps * ps, ky = r / ps, kx = r - ky * ps;
    const int y = (p / g) * ps + ky, x = (p % g) * ps + kx;
    v = px[(((size_t)b * 3 + c) * IS + y) * IS + x];
  }
  out[i] = (bf16_t)(pack2rt(v, 0.f, dt) & 0xffff);
}

// [class_embedding ; patch embeddings] + position_embedding -> pre_layrnorm -> fp32 residual stream (row = b*(np+1)+t)
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const float* __restrict__ patches, const float* __restrict__ cls_emb,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int rows, int T,
                                                           int cols, float* __restrict__ o32) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = cols >> 2, b = row / T, t = row - b * T;
  const float4* src = t == 0 ? (const float4*)cls_emb : (const float4*)(patches + ((size_t)b * (T - 1) + (t - 1)) * cols);
  const float4* pr = (const float4*)(pos + (size_t)t * cols);
  float4 v[MAX_V4];
#pragma unroll
  for (int i = 0; i < MAX_V4; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < n4) {
      const float4 a = src[c4], c = pr[c4];
      v[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    } else {
      v[i] = make_float4(0, 0, 0, 0);
    }
  }
  const RowStats st = row_stats(v, n4, lane, cols, eps);
  ln_store(v, st, gamma, beta, n4, lane, o32 + (size_t)row * cols, nullptr, 0);
}

// ---- PreFLMR attention fusion (rerank_model.py:276-319): additive attention bias over the cross-encoder tokens
// [query | image | context] from the retriever's raw score matrix scores[pair][S][Tq] (context token x query/image token):
//   adj[i < Tq][Tq + kc] = mult * softmax over kc of ts[kc][i]      (query row attends context)
//   adj[Tq + kc][j < Tq] = mult * softmax over j  of ts[kc][j]      (context row attends query)
//   0 elsewhere (the two self-attention blocks, and the padding columns up to ld), with ts[kc][*] = scores[2 + kc][*],
//   kc < Tc = S - ql.  One workgroup per pair.
// `row0` = first score row that belongs to the sequence (2 for the joint RerankModel sequence, 0 for the Interaction
// reranker, interaction_rerank_model.py:131-142, whose scores are [N, Lc, Lq] already).
__global__ __launch_bounds__(256) void fusion_adj_kernel(const float* __restrict__ scores, int S, int Tq, int Tc, float mult,
                                                         int pair0, float* __restrict__ adj, int ld, int row0) {
  extern __shared__ float colstat[];                 // [Tq] max, [Tq] 1/sum
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* ts = scores + ((size_t)(pair0 + b) * S + row0) * Tq;
  float* A = adj + (size_t)b * (Tq + Tc) * ld;
  for (int i = tid; i < Tq; i += 256) {              // column statistics (softmax over the context tokens)
    float mx = -INFINITY;
    for (int kc = 0; kc < Tc; ++kc) mx = fmaxf(mx, ts[(size_t)kc * Tq + i]);
    float sum = 0.f;
    for (int kc = 0; kc < Tc; ++kc) sum += expf(ts[(size_t)kc * Tq + i] - mx);
    colstat[i] = mx;
    colstat[Tq + i] = 1.f / sum;
  }
  __syncthreads();
  for (int i = wave; i < Tq; i += 4) {               // upper rows: [0 | softmax over context]
    float* row = A + (size_t)i * ld;
    const float mx = colstat[i], inv = colstat[Tq + i];
    for (int j = lane; j < ld; j += 64) {
      const int kc = j - Tq;
      row[j] = (kc >= 0 && kc < Tc) ? mult * (expf(ts[(size_t)kc * Tq + i] - mx) * inv) : 0.f;
    }
  }
  for (int kc = wave; kc < Tc; kc += 4) {            // lower rows: [softmax over query/image tokens | 0]
    const float* r = ts + (size_t)kc * Tq;
    float mx = -INFINITY;
    for (int j = lane; j < Tq; j += 64) mx = fmaxf(mx, r[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Tq; j += 64) sum += expf(r[j] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float* row = A + (size_t)(Tq + kc) * ld;
    for (int j = lane; j < ld; j += 64) row[j] = j < Tq ? mult * (expf(r[j] - mx) * inv) : 0.f;
  }
}

// ---- interaction rerankers: key bias over the concatenated [query tokens | context tokens] sequence from the
// retriever's 0/1 masks (interaction_rerank_model.py:153); also the two separate biases MORES needs.
__global__ void interaction_bias_kernel(const float* __restrict__ qmask, const float* __restrict__ cmask, int n, int Lq,
                                        int Lc, int pair_off, int K, float* __restrict__ cat_bias,
                                        float* __restrict__ q_bias, float* __restrict__ c_bias) {
  const int T = Lq + Lc;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * T) return;
  const int p = i / T, t = i - p * T;
  float b;
  if (t < Lq) {
    b = qmask[(size_t)((p + pair_off) / K) * Lq + t] != 0.f ? 0.f : -1e30f;
    if (q_bias) q_bias[(size_t)p * Lq + t] = b;
  } else {
    b = cmask[(size_t)p * Lc + (t - Lq)] != 0.f ? 0.f : -1e30f;
    if (c_bias) c_bias[(size_t)p * Lc + (t - Lq)] = b;
  }
  if (cat_bias) cat_bias[i] = b;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n4, int dt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = ((const float4*)x)[i];
  ((uint2*)y)[i] = make_uint2(pack2rt(a.x, a.y, dt), pack2rt(a.z, a.w, dt));
}

// rows (dst_batch, j<rows_take) <- src row (dst_batch + off)/bdiv - src_off : generic 16-byte row gather/broadcast
__global__ void gather_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n_dst_batches,
                                   int rows_take, int src_rows_per_batch, int row_u4, int batch_off, int bdiv,
                                   int src_batch_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_dst_batches * rows_take * row_u4;
  if (i >= total) return;
  const int c = (int)(i % row_u4);
  const size_t rj = i / row_u4;
  const int j = (int)(rj % rows_take), p = (int)(rj / rows_take);
  const int sb = (p + batch_off) / bdiv - src_batch_off;
  dst[i] = src[((size_t)sb * src_rows_per_batch + j) * row_u4 + c];
}

// ---- CLS classifier heads: logit_k[p] = <h32[p*T + 0, :], w_k> + b_k   (utils.py:101-108)
__global__ __launch_bounds__(256) void cls_heads_kernel(const float* __restrict__ h32, int T, int cols, int n_pairs,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ out1, float* __restrict__ out2) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const float4* x = (const float4*)(h32 + (size_t)p * T * cols);
  float a1 = 0.f, a2 = 0.f;
  for (int c4 = lane; c4 < (cols >> 2); c4 += 64) {
    const float4 v = x[c4], u1 = ((const float4*)w1)[c4], u2 = ((const float4*)w2)[c4];
    a1 += (v.x * u1.x + v.y * u1.y) + (v.z * u1.z + v.w * u1.w);
    a2 += (v.x * u2.x + v.y * u2.y) + (v.z * u2.z + v.w * u2.w);
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    out1[p] = a1 + b1[0];
    if (out2) out2[p] = a2 + b2[0];
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers
hipError_t rr_launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols,
                               float* out_f32, bf16_t* out_bf16, int dt, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || (cols & 3) || cols > 64 * 4 * MAX_V4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, gamma, beta, eps, rows, cols,
                     out_f32, out_bf16, dt, (float2*)nullptr);
  return hipGetLastError();
}

// LayerNorm that also leaves (mean, rstd) per row; out_f32 may be null (the fp32 stream is then recomputed where it
// is consumed, gemm_bf16.hip:ln_apply).
hipError_t rr_launch_layernorm_stats(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols,
                                     float* out_f32, bf16_t* out_bf16, float* stats_out, int dt, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || (cols & 3) || cols > 64 * 4 * MAX_V4 || !stats_out) return hipErrorInvalidValue;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, gamma, beta, eps, rows, cols,
                     out_f32, out_bf16, dt, (float2*)stats_out);
  return hipGetLastError();
}

// LayerNorm -> e4m3 rows + per-row scales (+ (mean, rstd)); cols % 4 == 0
hipError_t rr_launch_layernorm_q8(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols,
                                  uint8_t* out8, float* row_scale, float* stats_out, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || (cols & 3) || cols > 64 * 4 * MAX_V4 || !out8 || !row_scale) return hipErrorInvalidValue;
  hipLaunchKernelGGL(layernorm_q8_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, gamma, beta, eps, rows, cols, out8,
                     row_scale, (float2*)stats_out);
  return hipGetLastError();
}

// The residual VALUE the split-stream epilogue of gemm_kernel_hp forms from a (hi, lo) row pair: x = hi + lo, then — with
// statistics — (x - mean) * rstd * gamma + beta, the very expression (and operation order) of that epilogue, as fp32 rows.
// Test infrastructure for tests/test_gpu_ops.py: fed to the fp32-stream epilogue it gives a bit-exact expectation for the
// production split epilogues (x16 / lo_out), which no other kernel implements.
__global__ void split_residual_value_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo,
                                            const float2* __restrict__ stats, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, size_t rows, int cols, int dt, float* __restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= rows * (size_t)cols) return;
  const size_t r = i / cols;
  const int c = (int)(i - r * cols);
  const uint32_t hw = *(const uint32_t*)(hi + i), lw = *(const uint32_t*)(lo + i);
  const float2 h = dt ? unpack2<1>(hw) : unpack2<0>(hw), l = unpack2<1>(lw);
  float x0 = h.x + l.x, x1 = h.y + l.y;
  if (stats) {
    const float2 st = stats[r];
    x0 = (x0 - st.x) * st.y * gamma[c] + beta[c];
    x1 = (x1 - st.x) * st.y * gamma[c + 1] + beta[c + 1];
  }
  out[i] = x0;
  out[i + 1] = x1;
}
hipError_t rr_launch_split_residual_value(const bf16_t* hi, const bf16_t* lo, const float* stats, const float* gamma, const float* beta,
                                          int rows, int cols, int dt, float* out, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || (cols & 1) || !hi || !lo || !out || (stats && (!gamma || !beta))) return hipErrorInvalidValue;
  const size_t pairs = (size_t)rows * cols / 2;
  hipLaunchKernelGGL(split_residual_value_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, hi, lo, (const float2*)stats,
                     gamma, beta, (size_t)rows, cols, dt, out);
  return hipGetLastError();
}

hipError_t rr_launch_ln_finalize(const float* part, int nparts, int cols, float eps, int rows, float* stats, hipStream_t st,
                                 int* range_flag, float range_ss) {
  if (rows <= 0 || cols <= 0 || nparts != (cols + 127) / 128 || !part || !stats) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, (const float2*)part, nparts, cols, eps,
                     rows, (float2*)stats, range_flag, range_ss);
  return hipGetLastError();
}

hipError_t rr_launch_embed_ln(const int64_t* ids, const int64_t* tts, const float* word, const float* pos,
                              const float* type, const float* gamma, const float* beta, float eps, int rows, int S,
                              int cols, int vocab, int type_vocab, float* o32, bf16_t* o16, int dt, hipStream_t st) {
  if (rows <= 0 || (cols & 3) || cols > 64 * 4 * MAX_V4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, ids, tts, word, pos, type, gamma,
                     beta, eps, rows, S, cols, vocab, type_vocab, o32, o16, dt);
  return hipGetLastError();
}

hipError_t rr_launch_ce_embed_ln(const float* x, const float* pos, const float* type0, const float* gamma,
                                 const float* beta, float eps, int rows, int T, int cols, float* o32, bf16_t* o16,
                                 int dt, hipStream_t st, int s_text, int vis_pos0, int cls32_only) {
  if (rows <= 0 || (cols & 3) || cols > 64 * 4 * MAX_V4) return hipErrorInvalidValue;
  if (s_text < 0 || s_text > T) { s_text = T; vis_pos0 = T; }     // plain positions 0 .. T-1
  hipLaunchKernelGGL(ce_embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, pos, type0, gamma, beta, eps,
                     rows, T, cols, o32, o16, dt, s_text, vis_pos0, cls32_only);
  return hipGetLastError();
}

hipError_t rr_launch_li_normalize(const float* src, const int64_t* ids, int ids_stride, int n_pairs,
                                  int rows_per_batch, int D, int T, int t_off, int pair_off, int bdiv,
                                  int src_batch_off, bf16_t* dst, int dt, int normalize, const float* maskf, int split,
                                  int shift, hipStream_t st) {
  if (n_pairs <= 0 || rows_per_batch <= 0 || (D & 3) || D > 64 * 4 * MAX_V4 || bdiv <= 0) return hipErrorInvalidValue;
  const int rows = n_pairs * rows_per_batch;
  hipLaunchKernelGGL(li_normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, src, ids, ids_stride, n_pairs,
                     rows_per_batch, D, T, t_off, pair_off, bdiv, src_batch_off, dst, dt, normalize, maskf, split, shift);
  return hipGetLastError();
}

hipError_t rr_launch_key_bias(const int64_t* ids, const int64_t* am, int n, int S, int T, float* text_bias,
                              float* ce_bias, hipStream_t st) {
  const int total = n * T;
  hipLaunchKernelGGL(key_bias_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ids, am, n, S, T, text_bias,
                     ce_bias);
  return hipGetLastError();
}

hipError_t rr_launch_vit_im2col(const float* px, bf16_t* out, int B, int IS, int ps, int Kp, int dt, hipStream_t st) {
  const size_t total = (size_t)B * (IS / ps) * (IS / ps) * Kp;
  hipLaunchKernelGGL(vit_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, px, out, B, IS, ps, Kp, dt);
  return hipGetLastError();
}

hipError_t rr_launch_vit_embed_ln(const float* patches, const float* cls_emb, const float* pos, const float* gamma,
                                  const float* beta, float eps, int rows, int T, int cols, float* o32, hipStream_t st) {
  if (rows <= 0 || (cols & 3) || cols > 64 * 4 * MAX_V4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(vit_embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, patches, cls_emb, pos, gamma, beta, eps,
                     rows, T, cols, o32);
  return hipGetLastError();
}

hipError_t rr_launch_fusion_adj(const float* scores, int S, int Tq, int Tc, float mult, int pair0, int n, float* adj, int ld,
                                hipStream_t st, int row0) {
  if (n <= 0 || Tq <= 0 || Tc <= 0 || row0 < 0 || row0 + Tc > S || ld < Tq + Tc || (ld & 63) || Tq > 8192) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fusion_adj_kernel, dim3(n), dim3(256), 2 * Tq * sizeof(float), st, scores, S, Tq, Tc, mult, pair0, adj, ld,
                     row0);
  return hipGetLastError();
}

hipError_t rr_launch_joint_masks(const int64_t* ids, const int64_t* am, int n, int S, int P, int q_len,
                                 long long instruction_token, float* text_bias, float* li_mask, float* ce_bias,
                                 hipStream_t st) {
  hipLaunchKernelGGL(joint_masks_kernel, dim3((n + 3) / 4), dim3(256), 0, st, ids, am, n, S, P, q_len,
                     instruction_token, text_bias, li_mask, ce_bias);
  return hipGetLastError();
}

hipError_t rr_launch_interaction_bias(const float* qmask, const float* cmask, int n, int Lq, int Lc, int pair_off, int K,
                                      float* cat_bias, float* q_bias, float* c_bias, hipStream_t st) {
  const int total = n * (Lq + Lc);
  hipLaunchKernelGGL(interaction_bias_kernel, dim3((total + 255) / 256), dim3(256), 0, st, qmask, cmask, n, Lq, Lc,
                     pair_off, K, cat_bias, q_bias, c_bias);
  return hipGetLastError();
}

hipError_t rr_launch_f32_to_bf16(const float* x, bf16_t* y, size_t n, int dt, hipStream_t st) {
  if (n & 3) return hipErrorInvalidValue;
  const size_t n4 = n >> 2;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, y, n4, dt);
  return hipGetLastError();
}

hipError_t rr_launch_gather_rows(const void* src, void* dst, int n_dst_batches, int rows_take, int src_rows_per_batch,
                                 int row_bytes, int batch_off, int bdiv, int src_batch_off, hipStream_t st) {
  if (row_bytes & 15) return hipErrorInvalidValue;
  const size_t total = (size_t)n_dst_batches * rows_take * (row_bytes >> 4);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint4*)src,
                     (uint4*)dst, n_dst_batches, rows_take, src_rows_per_batch, row_bytes >> 4, batch_off, bdiv,
                     src_batch_off);
  return hipGetLastError();
}

hipError_t rr_launch_cls_heads(const float* h32, int T, int cols, int n_pairs, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* out1, float* out2, hipStream_t st) {
  hipLaunchKernelGGL(cls_heads_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, st, h32, T, cols, n_pairs, w1, b1, w2,
                     b2, out1, out2);
  return hipGetLastError();
}
