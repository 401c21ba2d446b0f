// Fused multi-head attention (head dim 64) for gfx950:  O = softmax(Q K^T + key_bias) V
//
// Restates HF BertSelfAttention's eager path as the reference calls it
// (/root/reference/src/models/flmr/models/flmr/modeling_flmr.py:1622 text encoder;
//  src/models/rerank/attention_fusion.py:133-144 cross encoder; modeling_flmr.py:640-658 mapping
//  network self/cross attention) without ever materialising the [N,heads,T,T] score tensor
// (1.26 GB at c3).  1/sqrt(dh) is folded into Wq at weight-pack time
// (0.125 is a power of two => bit-exact), so Q arrives pre-scaled.
//
// Structure: workgroup = 4 waves = 128 query rows of one (pair, head); wave = 32 query rows.
//   * S^T = K Q^T is issued "swapped" (A-operand = K rows, B-operand = Q rows) with
//     v_mfma_f32_32x32x16_bf16, so a lane owns ONE query column and 32 of the 64 keys of a tile:
//     the softmax row reductions are in-register plus one lane<->lane+32 exchange, and the fp32
//     accumulator tile is directly the B operand of the P·V product (cdna_hip_programming.md §3
//     "An accumulator tile as the next MFMA's operand") — P never touches LDS;
//   * O^T = V^T P^T: V is staged row-major [key][d] in LDS (coalesced from HBM) and consumed
//     column-major through ds_read_b64_tr_b16 (hardware transpose, T10);
//   * K/V tiles of 64 keys are double-buffered in LDS through registers (issue-early/write-late, T14):
//     the next tile's global loads are in flight under this tile's 16 MFMAs;
//   * key padding: additive fp32 bias per key in the log2 domain (0 valid, -1e30 masked, -inf beyond
//     Tk).  finfo.min-style semantics are preserved: a row with no valid key attends uniformly to
//     all Tk keys exactly like softmax over a constant row does in the reference.
#include "rr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int KT = 64;                     // keys per tile
constexpr int TILE_BYTES = KT * 64 * 2;    // 8 KiB (K or V tile)

// V image: 128-byte rows, 16-byte chunk index XORed with 4*((row>>1)&1): the 4 rows x 64 bytes a
// half-wave touches in one ds_read_b64_tr_b16 then cover all 64 banks once.
__device__ __forceinline__ int vswz(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4); }

__device__ __forceinline__ bf16x8 tr_pair(const char* vt, int key0, int d_chunk_off, int lane) {
  // two transposed 4x16 blocks: keys key0..key0+3 and key0+8..key0+11, columns by lane group.
  const int i = lane & 15, qd = i >> 2, p = i & 3;
  const int c = d_chunk_off + (p >> 1);
  const int o = (p & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(vt + vswz(key0 + qd, c) + o));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(vt + vswz(key0 + 8 + qd, c) + o));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int DT, bool DENSE, bool DIAG = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ q, int q_stride,
                                                       int q_batch_div, int q_batch_off,
                                                       const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, int kv_stride,
                                                       const float* __restrict__ key_bias, int heads,
                                                       int Tq, int Tk, bf16_t* __restrict__ out,
                                                       int out_stride, int groups,
                                                       const float* __restrict__ dense_bias, int dense_ld,
                                                       unsigned long long* __restrict__ stamps, int g_attn_prio) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES + 2 * KT * 4 + 16];
  char* const k_img = lds;                       // [2][8 KiB]
  char* const v_img = lds + 2 * TILE_BYTES;      // [2][8 KiB]
  float* const b_img = (float*)(lds + 4 * TILE_BYTES);  // [2][64] raw additive key bias
  int* const f_img = (int*)(lds + 4 * TILE_BYTES + 2 * KT * 4);   // [2] tile has a masked / out-of-range key

  // XCD-aware block map: workgroups go to the 8 XCDs round-robin by id, and all query blocks of one (sequence, head)
  // read the same K/V.  Keep them on ONE XCD (ids congruent mod 8, consecutive in dispatch order) so K/V are fetched
  // into that L2 once instead of once per query block (measured 5.6 GB beyond L2 per launch vs 2.5 GB algorithmic).
  const int nqb = (Tq + 127) >> 7;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = (local / nqb) * 8 + xcd, qblk = local - (local / nqb) * nqb;
  if (grp >= groups) return;
  const int b = grp / heads, head = grp - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int qrow = qblk * 128 + wave * 32 + (lane & 31);

  // ---- Q fragments (B operand of S^T = K Q^T): Q[query = lane&31][d = 16 i + 8 h + j]
  bf16x8 qf[4];
  {
    const bf16_t* qp = q + ((size_t)((b + q_batch_off) / q_batch_div) * Tq + min(qrow, Tq - 1)) * q_stride + head * 64 + 8 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) qf[i] = *(const bf16x8*)(qp + 16 * i);
  }

  // ---- K/V tile staging through registers: thread handles 16-byte chunks idx = tid, tid + 256
  const bf16_t* kbase = k + (size_t)b * Tk * kv_stride + head * 64;
  const bf16_t* vbase = v + (size_t)b * Tk * kv_stride + head * 64;
  uint4 kr0, kr1, vr0, vr1;
  float br = 0.f;
  const int srow0 = tid >> 3, srow1 = srow0 + 32, sc = tid & 7;   // chunk idx = tid, tid + 256
#define RR_LOAD_TILE(t)                                                                         \
  {                                                                                             \
    const size_t off0 = (size_t)min((t) * KT + srow0, Tk - 1) * kv_stride + sc * 8;             \
    const size_t off1 = (size_t)min((t) * KT + srow1, Tk - 1) * kv_stride + sc * 8;             \
    kr0 = *(const uint4*)(kbase + off0); kr1 = *(const uint4*)(kbase + off1);                   \
    vr0 = *(const uint4*)(vbase + off0); vr1 = *(const uint4*)(vbase + off1);                   \
    if (tid < KT) {                                                                             \
      const int key = (t) * KT + tid;                                                           \
      br = key < Tk ? (key_bias ? key_bias[(size_t)b * Tk + key] : 0.f) : -INFINITY;    \
    }                                                                                           \
  }
#define RR_WRITE_TILE(buf)                                                                      \
  {                                                                                             \
    *(uint4*)(k_img + (buf) * TILE_BYTES + swz128(srow0, sc)) = kr0;                            \
    *(uint4*)(k_img + (buf) * TILE_BYTES + swz128(srow1, sc)) = kr1;                            \
    *(uint4*)(v_img + (buf) * TILE_BYTES + vswz(srow0, sc)) = vr0;                              \
    *(uint4*)(v_img + (buf) * TILE_BYTES + vswz(srow1, sc)) = vr1;                              \
    if (tid < KT) {                                                                             \
      b_img[(buf) * KT + tid] = br;                                                             \
      const unsigned long long any = __ballot(br != 0.f);      /* wave 0 only: tid < 64 */     \
      if (tid == 0) f_img[buf] = any != 0ull;                                                   \
    }                                                                                           \
  }

  f32x16 o0, o1;   // O^T[d = 32*dblk + (r&3) + 8(r>>2) + 4h][query = lane&31]
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;   // running max (log2 domain) and this lane's partial row sum

  // diagnostic build: s_memtime marks per KV tile (read after the tile's barrier), summed per wave
  unsigned long long dg[5] = {0, 0, 0, 0, 0}, tmk[6];
#define RR_MARK(k) { if constexpr (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tmk[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } }
  const int nt = (Tk + KT - 1) / KT;
  RR_LOAD_TILE(0)
  RR_WRITE_TILE(0)
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    RR_MARK(0)
    if (t + 1 < nt) RR_LOAD_TILE(t + 1)
    const char* kt_ = k_img + buf * TILE_BYTES;
    const char* vt_ = v_img + buf * TILE_BYTES;
    const float* bt_ = b_img + buf * KT;

    // ---- S^T tile: keys 0..31 -> s0, 32..63 -> s1; reg r <-> key (r&3) + 8(r>>2) + 4h
    if (g_attn_prio) __builtin_amdgcn_s_setprio(2);   // MFMA sections outrank the softmax VALU of the co-resident waves
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 k0 = *(const bf16x8*)(kt_ + swz128(lane & 31, 2 * i + h));
      const bf16x8 k1 = *(const bf16x8*)(kt_ + swz128(32 + (lane & 31), 2 * i + h));
      s0 = mfma32<DT>(k0, qf[i], s0);
      s1 = mfma32<DT>(k1, qf[i], s1);
    }
    if (g_attn_prio) __builtin_amdgcn_s_setprio(0);
    RR_MARK(1)
    // ---- online softmax.  Running max m_run is kept in the RAW score domain; exponentials are exp2 of
    // LOG2E-scaled differences on the bare v_exp_f32 (arguments are <= 0, a flushed denormal is an exact 0 here).
    // DENSE: an additive bias per (query, key) on top of the per-key one (PreFLMR attention fusion,
    // attention_fusion.py:84-102): rows of dense_bias are [Tq][dense_ld], dense_ld a multiple of 64, zero padded
    const bool masked = DENSE || f_img[buf] != 0;    // wave-uniform: some key of this tile carries a bias
    float mx;
    if constexpr (DENSE) {
      const float* dp = dense_bias + ((size_t)b * Tq + min(qrow, Tq - 1)) * dense_ld + t * KT + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 d0 = *(const float4*)(dp + 8 * g);
        const float4 d1 = *(const float4*)(dp + 32 + 8 * g);
        s0[4 * g + 0] += d0.x; s0[4 * g + 1] += d0.y; s0[4 * g + 2] += d0.z; s0[4 * g + 3] += d0.w;
        s1[4 * g + 0] += d1.x; s1[4 * g + 1] += d1.y; s1[4 * g + 2] += d1.z; s1[4 * g + 3] += d1.w;
      }
    }
    if (masked) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b0 = *(const float4*)(bt_ + 8 * g + 4 * h);
        const float4 b1 = *(const float4*)(bt_ + 32 + 8 * g + 4 * h);
        s0[4 * g + 0] += b0.x; s0[4 * g + 1] += b0.y; s0[4 * g + 2] += b0.z; s0[4 * g + 3] += b0.w;
        s1[4 * g + 0] += b1.x; s1[4 * g + 1] += b1.y; s1[4 * g + 2] += b1.z; s1[4 * g + 3] += b1.w;
      }
    }
    // 32-way max; this file is built with -fno-honor-nans -mno-amdgpu-ieee (build.py) so that fmaxf lowers to bare
    // v_max3_f32 — in IEEE mode hipcc canonicalises every MFMA output first (+32 VALU per tile).  No NaN can occur:
    // inputs are finite and the only non-finite values are the -inf biases of out-of-range keys.
    mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);            // finite: every tile has >= 1 in-range key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);   // first tile: exp2(-inf) = 0
    m_run = m_new;
    if (masked) {
      // (s + bias) - m is exactly 0 for a fully masked row (all entries -1e30): uniform attention, as the
      // reference's finfo.min mask gives; a fused multiply-add form would not cancel exactly.
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f((s0[r] - m_new) * LOG2E);
        s1[r] = __builtin_amdgcn_exp2f((s1[r] - m_new) * LOG2E);
      }
    } else {
      const float c = -m_new * LOG2E;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], LOG2E, c));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], LOG2E, c));
      }
    }
    // row sum over this lane's 32 probabilities, written as a tree so the adds pair up (v_pk_add_f32)
    typedef __attribute__((ext_vector_type(2))) float f32x2v;
    f32x2v acc2 = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      acc2 += f32x2v{s0[r], s0[r + 1]};
      acc2 += f32x2v{s1[r], s1[r + 1]};
    }
    const float ps = acc2[0] + acc2[1];
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

    if (g_attn_prio) __builtin_amdgcn_s_setprio(2);
    RR_MARK(2)
    // ---- O^T += V^T P^T.  P fragment for k-step s of key block kb = regs 8s..8s+7 (k order:
    // element j <-> key 16 s + 8 (j>>2) + 4 h + (j&3)); V fragment gathers the same keys.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        u32x4 pw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = kb == 0 ? s0[8 * s + 2 * j] : s1[8 * s + 2 * j];
          const float hi = kb == 0 ? s0[8 * s + 2 * j + 1] : s1[8 * s + 2 * j + 1];
          pw[j] = pack2<DT>(lo, hi);
        }
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int key0 = kb * 32 + 16 * s + 4 * h;
        const int cg = 2 * ((lane >> 4) & 1);      // 16-lane group -> d columns 16*(g&1) within the 32-d block
        const bf16x8 v0 = tr_pair(vt_, key0, 0 + cg, lane);   // d block 0: chunks 0..3
        const bf16x8 v1 = tr_pair(vt_, key0, 4 + cg, lane);   // d block 1: chunks 4..7
        o0 = mfma32<DT>(v0, pf, o0);
        o1 = mfma32<DT>(v1, pf, o1);
      }
    }
    if (g_attn_prio) __builtin_amdgcn_s_setprio(0);
    RR_MARK(3)
    if (t + 1 < nt) RR_WRITE_TILE(buf ^ 1)
    RR_MARK(4)
    __syncthreads();
    RR_MARK(5)
    if constexpr (DIAG) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k_ = 0; k_ < 5; ++k_) dg[k_] += tmk[k_ + 1] - tmk[k_];
    }
  }
  if constexpr (DIAG) {
    if (stamps && lane == 0)
      for (int k_ = 0; k_ < 5; ++k_) stamps[((size_t)blockIdx.x * 4 + wave) * 8 + k_] = dg[k_];
    if (stamps && tid == 0) stamps[((size_t)blockIdx.x * 4) * 8 + 7] = (unsigned long long)nt;
  }
#undef RR_MARK

  // ---- epilogue: O = O^T / l ; lane writes 4 consecutive d per register group
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < Tq) {
    bf16_t* op = out + ((size_t)b * Tq + qrow) * out_stride + head * 64 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(uint2*)(op + 8 * g) = make_uint2(pack2<DT>(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                         pack2<DT>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
      *(uint2*)(op + 32 + 8 * g) = make_uint2(pack2<DT>(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                              pack2<DT>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
    }
  }
}

}  // namespace

static unsigned long long* g_attn_stamps = nullptr;
static int g_attn_prio_host = 1;   // rr_set_tuning("attn_prio"): MFMA sections at wave priority 2, softmax at 0 (+4 % attention)
extern "C" int rr_set_attn_prio(int on) { g_attn_prio_host = on; return 0; }
extern "C" int rr_set_attn_stamps(void* device_buf) {   // diagnostic: 4 waves x 8 uint64 per workgroup, or NULL
  g_attn_stamps = (unsigned long long*)device_buf;
  return 0;
}

hipError_t rr_launch_attention(const bf16_t* q, int q_stride, int q_batch_div, int q_batch_off, const bf16_t* k,
                               const bf16_t* v, int kv_stride, const float* key_bias, int B, int heads,
                               int Tq, int Tk, bf16_t* out, int out_stride, int dt, hipStream_t st,
                               const float* dense_bias, int dense_ld) {
  if (dt != 0 && dt != 1) return hipErrorInvalidValue;
  if (B <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0 || q_batch_div <= 0 || q_batch_off < 0) return hipErrorInvalidValue;
  if ((q_stride & 7) || (kv_stride & 7) || (out_stride & 3)) return hipErrorInvalidValue;
  const long groups = (long)B * heads, nqb = (Tq + 127) / 128, nblk = ((groups + 7) / 8) * 8 * nqb;
  if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nblk), block(256);
  if (dense_bias && (dense_ld < Tk || (dense_ld & 63))) return hipErrorInvalidValue;
#define RR_ATTN(DT_, DENSE_)                                                                                        \
  hipLaunchKernelGGL((attn_fwd_kernel<DT_, DENSE_>), grid, block, 0, st, q, q_stride, q_batch_div, q_batch_off, k, v, \
                     kv_stride, key_bias, heads, Tq, Tk, out, out_stride, (int)groups, dense_bias, dense_ld, nullptr, g_attn_prio_host)
  if (g_attn_stamps && dt == 0 && !dense_bias) {   // diagnostic timeline (tools/attn_timeline.py)
    hipLaunchKernelGGL((attn_fwd_kernel<0, false, true>), grid, block, 0, st, q, q_stride, q_batch_div, q_batch_off, k, v,
                       kv_stride, key_bias, heads, Tq, Tk, out, out_stride, (int)groups, dense_bias, dense_ld, g_attn_stamps, g_attn_prio_host);
    return hipGetLastError();
  }
  if (dt == 0) { if (dense_bias) RR_ATTN(0, true); else RR_ATTN(0, false); }
  else { if (dense_bias) RR_ATTN(1, true); else RR_ATTN(1, false); }
#undef RR_ATTN
  return hipGetLastError();
}
