// Fused multi-head attention (head dim 64) for gfx950:  O = softmax(Q K^T + key_bias) V
//
// Restates HF BertSelfAttention's eager path as the reference calls it
// (/root/reference/src/models/flmr/models/flmr/modeling_flmr.py:1622 text encoder;
//  src/models/rerank/attention_fusion.py:133-144 cross encoder; modeling_flmr.py:640-658 mapping
//  network self/cross attention) without ever materialising the [N,heads,T,T] score tensor
// (1.26 GB at c3).  log2(e)/sqrt(dh) is folded into Wq and bq at weight-pack time, so Q arrives
// pre-scaled and Q K^T is already the argument of the hardware's base-2 exponential: no multiply per score.  Additive
// biases (-1e30 / -inf per key) need no rescaling; the dense per-(query, key) bias is multiplied by log2(e) when added.
//
// Structure: workgroup = 4 waves = 128 query rows of one (pair, head); wave = 32 query rows.
//   * S^T = K Q^T is issued "swapped" (A-operand = K rows, B-operand = Q rows) with
//     v_mfma_f32_32x32x16_bf16, so a lane owns ONE query column and 32 of the 64 keys of a tile:
//     the softmax row reductions are in-register plus one lane<->lane+32 exchange, and the fp32
//     accumulator tile is directly the B operand of the P·V product (cdna_hip_programming.md §3
//     "An accumulator tile as the next MFMA's operand") — P never touches LDS;
//   * O^T = V^T P^T: V is staged row-major [key][d] in LDS (coalesced from HBM) and consumed
//     column-major through ds_read_b64_tr_b16 (hardware transpose, T10);
//   * K/V tiles of 64 keys are double-buffered in LDS and arrive by LDS-DMA (global_load_lds_dwordx4, no staging
//     registers; both swizzles applied on the global side): the next tile's 16 pieces are in flight under this tile's
//     16 MFMAs;
//   * key padding: additive fp32 bias per key (0 valid, -1e30 masked, -inf beyond Tk).  finfo.min-style semantics are
//     preserved: a row with no valid key attends uniformly to all Tk keys exactly like softmax over a constant row
//     does in the reference;
//   * two schedules over the same tile code (attn_block): the online softmax, exact for any input, and the
//     fixed-reference form (no running maximum, no O rescale; 128 VGPRs = 4 waves per SIMD) whose rare failures — a
//     row sum beyond 2^64, a sequence without a valid key — are flagged per workgroup and recomputed online by a
//     second, normally empty launch.
#include <map>
#include <mutex>
#include <utility>

#include "rr_common.h"

// Row sum of the 64-row form (per tile and sub-block: the lane's 32 probabilities).  1 (default since round 4) = four independent
// v_add_f32 chains; 0 = round 1-3's single chain of v_pk_add_f32, which hipcc emits as 16 DEPENDENT packed adds with an `s_nop 0`
// behind each; 2 = a tree of packed adds.  Same box, alternating processes (profiles/r04_s_attn_rowsum_ab.log): 0.887 / 0.875 /
// 0.875 ms per launch (fp16), 0.855 / 0.842 / 0.842 (bf16).  The scalar form also removes the only packed-f32 arithmetic fed by
// MFMA results from this file (build.py, the packed-f32 hazard note).
#ifndef RR_ATTN_ROWSUM
#define RR_ATTN_ROWSUM 1
#endif

namespace {
// The lane's 32 probabilities of one 32-query x 64-key score block, summed in ONE order for every schedule (the 32- and the 64-row
// fixed-reference forms are bit-for-bit equal, tests/test_gpu_ops.py).
__device__ __forceinline__ float row_sum32(const f32x16& a, const f32x16& b) {
#if RR_ATTN_ROWSUM != 1
  typedef __attribute__((ext_vector_type(2))) float f32x2v;
#endif
#if RR_ATTN_ROWSUM == 1       // four independent scalar chains
  float c4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c4[j] = a[j] + b[j];
#pragma unroll
  for (int r = 4; r < 16; r += 4)
#pragma unroll
    for (int j = 0; j < 4; ++j) { c4[j] += a[r + j]; c4[j] += b[r + j]; }
  return (c4[0] + c4[1]) + (c4[2] + c4[3]);
#elif RR_ATTN_ROWSUM == 2     // a tree of packed adds: no add waits for the one before it
  f32x2v t8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t8[i] = f32x2v{a[2 * i], a[2 * i + 1]} + f32x2v{b[2 * i], b[2 * i + 1]};
#pragma unroll
  for (int i = 0; i < 4; ++i) t8[i] += t8[i + 4];
  t8[0] += t8[2]; t8[1] += t8[3];
  t8[0] += t8[1];
  return t8[0][0] + t8[0][1];
#else                         // rounds 1-3: one chain of packed adds
  f32x2v acc2 = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    acc2 += f32x2v{a[r], a[r + 1]};
    acc2 += f32x2v{b[r], b[r + 1]};
  }
  return acc2[0] + acc2[1];
#endif
}
}  // namespace

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr float LOG2E = 1.4426950408889634f;   // only the DENSE bias is still scaled in the kernel

// -x rounded to the 16-bit operand type: the fixed reference of a row as an MFMA operand (bits) and as the float it denotes
template <int DT>
__device__ __forceinline__ uint32_t ref16(float x, float& back) {
  const uint32_t u = pack2<DT>(x, 0.f) & 0xffffu;
  if constexpr (DT == 0) back = __builtin_bit_cast(float, u << 16);
  else back = (float)__builtin_bit_cast(_Float16, (unsigned short)u);
  return u;
}
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

// a wave-uniform pointer the compiler cannot prove uniform: pin it to SGPRs (the LDS-DMA takes its base as an "s" operand)
__device__ __forceinline__ const bf16_t* uniform_ptr(const bf16_t* p) {
  const unsigned long long u = (unsigned long long)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const bf16_t*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

// Key bias: a chunk of BIAS_TILES tiles (1024 keys) is resident in LDS with one flag word per tile, loaded by the whole
// workgroup before the tile loop (and again every 16 tiles for longer sequences): the tile loop itself holds no global load
// the compiler can see.  With one in it, hipcc's scoreboard — which cannot see the hand-counted LDS-DMA — guarded the load's
// destination register with s_waitcnt vmcnt(0) right behind the four DMA pieces just issued, i.e. every tile waited for its
// own prefetch to land (the QK^T section of the timeline: 2330 cycles against 260 of MFMA).
constexpr int BIAS_TILES = 16;
constexpr int ATTN_LDS_BYTES = 4 * TILE_BYTES + BIAS_TILES * KT * 4 + BIAS_TILES * 4;   // K/V double buffers, key bias chunk, tile flags

// Load bias chunk c (keys [1024 c, 1024 c + 1024) of sequence b; 0 without a bias, -inf beyond Tk) and its tile flags
// (bit 0: some key of the tile has a bias, bit 1: no key of the tile is valid).  All 256 threads; two barriers inside; the
// caller guarantees that no wave still reads the previous chunk.
__device__ __forceinline__ void load_bias_chunk(float* const b_all, int* const f_all, const float* key_bias, const int b,
                                                const int Tk, const int c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < BIAS_TILES * KT / 256; ++j) {
    const int key = c * (BIAS_TILES * KT) + j * 256 + tid;
    b_all[j * 256 + tid] = key < Tk ? (key_bias ? key_bias[(size_t)b * Tk + key] : 0.f) : -INFINITY;
  }
  __syncthreads();
#pragma unroll
  for (int tile = wave; tile < BIAS_TILES; tile += 4) {
    const float v = b_all[tile * KT + lane];
    const unsigned long long any = __ballot(v != 0.f), valid = __ballot(v > -1e29f);
    if (lane == 0) f_all[tile] = (any != 0ull ? 1 : 0) | (valid == 0ull ? 2 : 0);
  }
  __syncthreads();
}

// The 16 tile flags of the resident chunk as ONE scalar word per wave (bit i: tile i has a bias; bit 16 + i: tile i has no valid
// key), read once behind load_bias_chunk: the tile loop then tests a bit instead of reading its flag word from LDS and waiting
// for lgkmcnt(0) at the top of every tile.
__device__ __forceinline__ uint32_t chunk_flag_bits(const int* const f_all) {
  const int f = f_all[threadIdx.x & (BIAS_TILES - 1)];
  const uint32_t any = (uint32_t)__ballot((f & 1) != 0) & 0xFFFFu, none = (uint32_t)__ballot((f & 2) != 0) & 0xFFFFu;
  return any | (none << 16);
}
__device__ __forceinline__ int tile_flags_of(const uint32_t bits, const int t) {
  const int i = t & (BIAS_TILES - 1);
  return (int)((bits >> i) & 1u) | (int)(((bits >> (16 + i)) & 1u) << 1);
}

// ---- output rows through a wave-private LDS slice.  The accumulators hold O^T (lane = query row, registers = head
// dimensions), so storing them directly is 16 instructions of 8 bytes per lane that each touch 64 different rows.  Staged
// as rows [query][64 d] in the K/V buffers, which are free after the last tile's barrier, a wave writes its rows as 16
// bytes per lane, 8 lanes per 128-byte row, 8 rows per instruction (a quarter of the store instructions, whole lines).
// Same-wave LDS accesses execute in issue order: no barrier between the two halves.
// Image: 128-byte rows, 16-byte chunk c of row r at chunk slot c ^ (r & 7), and in rows with bit 3 set the two 8-byte
// halves of every chunk swapped.  Conflict-free both ways: a ds_write_b64 lane group is 16 rows of one chunk column and
// one half — the XOR spreads rows r..r+7 over the eight chunk slots, the half swap keeps rows r and r + 8 on different
// banks — and a ds_read_b128 lane group (four runs of four lanes in four different rows) meets four different 64-byte
// quarters of the 256-byte bank row.  (The first version, a plain 144-byte pitch, was 2-way on the writes and on the
// reads: SQ_LDS_BANK_CONFLICT 12.5 % of this kernel's LDS cycles in profiles/r02_z_sq_counters.json.)
constexpr int O_PITCH = 128;
static_assert(4 * 64 * O_PITCH <= ATTN_LDS_BYTES, "four waves x 64 output rows must fit the workgroup's LDS");
template <int DT>
__device__ __forceinline__ void stage_o_rows(char* slice, int row, int h, const f32x16& o0, const f32x16& o1, float inv) {
  char* rp = slice + row * O_PITCH + ((8 * h) ^ (row & 8));
  const int x = row & 7;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *(uint2*)(rp + ((g ^ x) << 4)) = make_uint2(pack2<DT>(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2<DT>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
    *(uint2*)(rp + (((4 + g) ^ x) << 4)) = make_uint2(pack2<DT>(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2<DT>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
  }
}
template <int ROWS>
__device__ __forceinline__ void store_o_rows(const char* slice, int lane, bf16_t* out_rows /* row 0 of the slice, head offset applied */,
                                             int out_stride, int rows_valid) {
#pragma unroll
  for (int u = 0; u < ROWS / 8; ++u) {
    const int i = lane + 64 * u, row = i >> 3, c = i & 7;
    uint4 v = *(const uint4*)(slice + row * O_PITCH + ((c ^ (row & 7)) << 4));
    if (u & 1) v = make_uint4(v.z, v.w, v.x, v.y);            // row = lane/8 + 8u: bit 3 of the row is bit 0 of u
    if (row < rows_valid) *(uint4*)(out_rows + (size_t)row * out_stride + c * 8) = v;
  }
}

struct AttnArgs {
  const bf16_t* q; int q_stride, q_batch_div, q_batch_off;
  const bf16_t* k; const bf16_t* v; int kv_stride;
  const float* key_bias; int heads, Tq, Tk;
  bf16_t* out; int out_stride, groups;
  const float* dense_bias; int dense_ld;
  unsigned long long* stamps; int tuning;
  int* flags; int nblk;      // fixed-reference schedule: one word per workgroup of the grid, 1 = recompute online
  unsigned long long* redo_stats;   // diagnostic (rr_set_attn_redo_stats): [0] += flagged workgroups, [1] += workgroups looked at; nullptr in production
  int rev;                          // 1: the grid walks the (sequence, head) groups from the last to the first (rr_m_direction_next)
};

// XCD-aware block map: workgroups go to the 8 XCDs round-robin by id, and all query blocks of one (sequence, head)
// read the same K/V.  Keep them on ONE XCD (ids congruent mod 8, consecutive in dispatch order) so K/V are fetched
// into that L2 once instead of once per query block (measured 5.6 GB beyond L2 per launch vs 2.5 GB algorithmic).
// XCD x takes the x-th CONTIGUOUS eighth of the (sequence, head) groups — the rows the x-th XCD of the persistent GEMM before and
// behind this launch wrote / will read (its tile lists are chunked the same way) — from its first to its last group, or from the
// last to the first (rev: rr_m_direction_next), so that with alternating directions each XCD starts on the rows that were
// written last.  Which XCD computes a group changes nothing in its values.  bid must be wave-uniform; nqb = query blocks per
// (sequence, head).  False for the padding ids (the grid holds ceil(groups / 8) slots per XCD).
__device__ __forceinline__ bool block_map(const int bid, const int nqb, const int groups, int& grp, int& qblk, const int rev = 0) {
  const int xcd = bid & 7, local = bid >> 3;
  const int j = local / nqb;                                // this XCD's j-th group, in dispatch order
  qblk = local - j * nqb;
  const int q8 = groups >> 3, r8 = groups & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, chunk_n = q8 + (xcd < r8 ? 1 : 0);
  grp = chunk0 + (rev ? chunk_n - 1 - j : j);
  return j < chunk_n;
}

// One workgroup's share: the 128 query rows [128 qblk, 128 qblk + 128) of (pair, head) = grp.
//   FIXED = false: online softmax (running max, O rescaled by exp(m_old - m_new) every tile) — exact for any input.
//   FIXED = true : the exponentials are taken against a FIXED per-row reference (the row maximum of the first tile that
//     holds a valid key) and the key bias is the initial value of the QK^T accumulators: no running maximum, no bias add,
//     no O rescale, and the reference itself is added by the matrix core: one more link in the QK^T chain whose K-side
//     fragment is the constant 1 and whose Q-side fragment holds the (16-bit rounded) reference, so a score leaves the
//     MFMA as the exponent and the VALU does the bare v_exp_f32, the row-sum add and the 16-bit pack;
//     with LDS-DMA staging 128 VGPRs, i.e. 4 waves per SIMD.  softmax is shift-invariant and fp32 keeps its relative precision at any exponent, so
//     the result differs from the online form in rounding only, as long as nothing overflows.  Tiles without a valid
//     key are skipped.  Returns true when the schedule did not hold: a row sum left 2^64 (6e4 with fp16 operands; a
//     later score far above the reference — inf and NaN fail the test too), or the sequence has no valid key at all (a
//     fully masked row must come out uniform, see the header); the caller then recomputes the workgroup online.
template <int DT, bool DENSE, bool DIAG, bool FIXED>
__device__ __forceinline__ bool attn_block(const int grp, const int qblk, char* const lds, const AttnArgs& a) {
  char* const k_img = lds;                       // [2][8 KiB]
  char* const v_img = lds + 2 * TILE_BYTES;      // [2][8 KiB]
  float* const b_all = (float*)(lds + 4 * TILE_BYTES);                        // [16][64] additive key bias of the chunk
  int* const f_all = (int*)(lds + 4 * TILE_BYTES + BIAS_TILES * KT * 4);      // [16] tile flags
  const int Tq = a.Tq, Tk = a.Tk, kv_stride = a.kv_stride;
  const float* const key_bias = a.key_bias;

  const int b = grp / a.heads, head = grp - b * a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int qrow = qblk * 128 + wave * 32 + (lane & 31);
  unsigned long long t_begin = 0;
  if constexpr (DIAG) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin) :: "memory");

  // ---- Q fragments (B operand of S^T = K Q^T): Q[query = lane&31][d = 16 i + 8 h + j]
  bf16x8 qf[4];
  {
    const bf16_t* qp = a.q + ((size_t)((b + a.q_batch_off) / a.q_batch_div) * Tq + min(qrow, Tq - 1)) * a.q_stride + head * 64 + 8 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) qf[i] = *(const bf16x8*)(qp + 16 * i);
  }

  // ---- K/V tile staging by LDS-DMA (no staging registers): a tile is 8 + 8 pieces of 1 KiB = 8 rows x 128 B; wave w
  // moves rows 16w..16w+15 of K and of V (2 + 2 pieces).  The DMA writes lane i's 16 bytes at piece + 16 i, i.e. row
  // i>>3, slot i&7, so the swizzles of the two images are applied on the global side: the lane fetches the chunk that
  // belongs in its slot.  Rows beyond Tk re-read row Tk-1 (their scores get a -inf bias).
  const bf16_t* kbase = uniform_ptr(a.k + (size_t)b * Tk * kv_stride + head * 64);
  const bf16_t* vbase = uniform_ptr(a.v + (size_t)b * Tk * kv_stride + head * 64);
  const uint32_t stride2 = (uint32_t)kv_stride * 2u;
  const int r0 = wave * 16 + (lane >> 3);                                   // piece 0 row; piece 1: + 8
  const uint32_t ck0 = (uint32_t)((lane & 7) ^ (lane >> 4)) << 4;           // swz128: chunk ^ ((row >> 1) & 7); piece 1: ^ 4
  const uint32_t cv = (uint32_t)((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4;   // vswz: chunk ^ (((row >> 1) & 1) << 2)
  const uint32_t dst0 = __builtin_amdgcn_readfirstlane(lds_addr(k_img) + wave * 2048);
#define RR_LOAD_TILE(t, buf)                                                                    \
  {                                                                                             \
    const uint32_t ro0 = (uint32_t)min((t) * KT + r0, Tk - 1) * stride2;                        \
    const uint32_t ro1 = (uint32_t)min((t) * KT + r0 + 8, Tk - 1) * stride2;                    \
    const uint32_t kd = dst0 + (buf) * TILE_BYTES;                                              \
    glds16_so(kbase, ro0 + ck0, kd);                                                            \
    glds16_so(kbase, ro1 + (ck0 ^ 64u), kd + 1024);                                             \
    glds16_so(vbase, ro0 + cv, kd + 2 * TILE_BYTES);                                            \
    glds16_so(vbase, ro1 + cv, kd + 2 * TILE_BYTES + 1024);                                     \
  }
#define RR_WRITE_TILE(buf) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* this wave's four pieces have landed */

  f32x16 o0, o1;   // O^T[d = 32*dblk + (r&3) + 8(r>>2) + 4h][query = lane&31]
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;   // online: running max (raw score domain); this lane's partial row sum
  bool need_ref = true;                   // fixed reference: minus the row maximum of the first tile with a valid key
                                          // (all rows of a workgroup see the same keys, so this flips for all lanes at once)
  // the reference as one more k-step of S^T = K Q^T: K side = 1 at k = 0, Q side = reference at k = 0 (lanes with h = 0)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
  // (both fragments are rebuilt from one register each per tile: at 128 VGPRs two resident quads would spill)
  uint32_t kone0 = h == 0 ? (DT == 0 ? 0x3F80u : 0x3C00u) : 0u, qc0 = 0u;

  // diagnostic build: s_memtime marks per KV tile (read after the tile's barrier), summed per wave
  unsigned long long dg[5] = {0, 0, 0, 0, 0}, tmk[6];
#define RR_MARK(k) { if constexpr (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tmk[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } }
  const int nt = (Tk + KT - 1) / KT;
  const bool prio = (a.tuning & 1) != 0;
  RR_LOAD_TILE(0, 0)
  load_bias_chunk(b_all, f_all, key_bias, b, Tk, 0);
  uint32_t flag_bits = chunk_flag_bits(f_all);
  RR_WRITE_TILE(0)
  // A wait the compiler can see: the LDS-DMA is counted by hand (asm), so without it hipcc's scoreboard still holds the Q
  // loads as "in flight" on every trip of the loop and puts s_waitcnt vmcnt(6..3) in front of the QK^T MFMAs — which, with
  // the next tile's four DMA pieces just issued, stalls the chain until two of them have LANDED.
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt / lgkmcnt unconstrained
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (__builtin_expect(t != 0 && (t & (BIAS_TILES - 1)) == 0, 0)) {      // sequences beyond 1024 keys: next bias chunk
      load_bias_chunk(b_all, f_all, key_bias, b, Tk, t / BIAS_TILES);
      flag_bits = chunk_flag_bits(f_all);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    RR_MARK(0)
    if (t + 1 < nt) RR_LOAD_TILE(t + 1, buf ^ 1)
    const char* kt_ = k_img + buf * TILE_BYTES;
    const char* vt_ = v_img + buf * TILE_BYTES;
    const float* bt_ = b_all + (t & (BIAS_TILES - 1)) * KT;
    // wave-uniform: some key of this tile carries a bias.  DENSE: an additive bias per (query, key) on top of the
    // per-key one (PreFLMR attention fusion, attention_fusion.py:84-102): rows of dense_bias are [Tq][dense_ld],
    // dense_ld a multiple of 64, zero padded; it only exists in the online form.
    const int tile_flags = tile_flags_of(flag_bits, t);   // bit 0: some key has a bias; bit 1: no valid key
    const bool masked = DENSE || (tile_flags & 1);
    // Fixed-reference form only: a tile without a single valid key (tail padding) adds exactly 0 to every row sum and to
    // O, so its QK^T, softmax and P.V are skipped (the staging of the next tile and the barrier
    // stay).  If NO tile has a valid key the workgroup ends without a reference and is recomputed online, where every
    // tile is processed: that is the case in which masked keys do count (uniform attention).
    if (!(FIXED && (tile_flags & 2))) {

    // ---- S^T tile: keys 0..31 -> s0, 32..63 -> s1; reg r <-> key (r&3) + 8(r>>2) + 4h
    f32x16 s0, s1;
    auto qk = [&]() __attribute__((always_inline)) {
      if (prio) __builtin_amdgcn_s_setprio(2);   // MFMA sections outrank the softmax VALU of the co-resident waves
      if constexpr (FIXED) {     // + reference: register operands only, so it runs under the first K fragments' LDS latency
        asm volatile("" : "+v"(kone0), "+v"(qc0));
        const bf16x8 kone = __builtin_bit_cast(bf16x8, u32x4_{kone0, 0u, 0u, 0u});
        const bf16x8 qc = __builtin_bit_cast(bf16x8, u32x4_{qc0, 0u, 0u, 0u});
        s0 = mfma32<DT>(kone, qc, s0);
        s1 = mfma32<DT>(kone, qc, s1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 k0 = *(const bf16x8*)(kt_ + swz128(lane & 31, 2 * i + h));
        const bf16x8 k1 = *(const bf16x8*)(kt_ + swz128(32 + (lane & 31), 2 * i + h));
        s0 = mfma32<DT>(k0, qf[i], s0);
        s1 = mfma32<DT>(k1, qf[i], s1);
      }
      if (prio) __builtin_amdgcn_s_setprio(0);
    };
    if (FIXED && masked) {      // the key bias is the accumulators' initial value: straight from LDS, no VALU
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b0 = *(const float4*)(bt_ + 8 * g + 4 * h);
        const float4 b1 = *(const float4*)(bt_ + 32 + 8 * g + 4 * h);
        s0[4 * g + 0] = b0.x; s0[4 * g + 1] = b0.y; s0[4 * g + 2] = b0.z; s0[4 * g + 3] = b0.w;
        s1[4 * g + 0] = b1.x; s1[4 * g + 1] = b1.y; s1[4 * g + 2] = b1.z; s1[4 * g + 3] = b1.w;
      }
      qk();
    } else {                    // separate copy of the MFMA chain: its first link takes the inline constant 0
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      qk();
    }
    RR_MARK(1)
    // 32-way max; this file is built with -fno-honor-nans -mno-amdgpu-ieee (build.py) so that fmaxf lowers to bare
    // v_max3_f32 — in IEEE mode hipcc canonicalises every MFMA output first (+32 VALU per tile).  No NaN can occur:
    // inputs are finite and the only non-finite values are the -inf biases of out-of-range keys.
    auto row_max = [&]() __attribute__((always_inline)) {
      float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
      return fmaxf(mx, __shfl_xor(mx, 32, 64));
    };
    if constexpr (FIXED) {
      if (__builtin_expect(need_ref, 0)) {        // first tile(s) only: take the reference from the first valid keys
        const float mx = row_max();
        if (mx > -1e29f) {
          float c_ref;
          const uint32_t c16 = ref16<DT>(-mx, c_ref);     // any reference near the maximum does: the rounded one is used throughout
          qc0 = h == 0 ? c16 : 0u;
          need_ref = false;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s0[r] += c_ref; s1[r] += c_ref; }   // this tile's chain ran without it
        }
      }
      // the bare v_exp_f32 per score: a masked key is (-1e30 + q.k + ref) -> 0, a key beyond Tk is -inf -> 0
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r]);
        s1[r] = __builtin_amdgcn_exp2f(s1[r]);
      }
    } else {
      // ---- online softmax.  Scores and the running max m_run are in the log2 domain (Q is pre-scaled); exponentials
      // are the bare v_exp_f32 of differences (arguments are <= 0, a flushed denormal is an exact 0 here).
      if constexpr (DENSE) {
        const float* dp = a.dense_bias + ((size_t)b * Tq + min(qrow, Tq - 1)) * a.dense_ld + t * KT + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 d0 = *(const float4*)(dp + 8 * g);
          const float4 d1 = *(const float4*)(dp + 32 + 8 * g);
          s0[4 * g + 0] = fmaf(d0.x, LOG2E, s0[4 * g + 0]); s0[4 * g + 1] = fmaf(d0.y, LOG2E, s0[4 * g + 1]);
          s0[4 * g + 2] = fmaf(d0.z, LOG2E, s0[4 * g + 2]); s0[4 * g + 3] = fmaf(d0.w, LOG2E, s0[4 * g + 3]);
          s1[4 * g + 0] = fmaf(d1.x, LOG2E, s1[4 * g + 0]); s1[4 * g + 1] = fmaf(d1.y, LOG2E, s1[4 * g + 1]);
          s1[4 * g + 2] = fmaf(d1.z, LOG2E, s1[4 * g + 2]); s1[4 * g + 3] = fmaf(d1.w, LOG2E, s1[4 * g + 3]);
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
      const float m_new = fmaxf(m_run, row_max());     // finite: every tile has >= 1 in-range key
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
      m_run = m_new;
      // (s + bias) - m is exactly 0 for a fully masked row (all entries -1e30): uniform attention, as the reference's
      // finfo.min mask gives
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
      }
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    l_run += row_sum32(s0, s1);       // one summation order for every schedule

    if (prio) __builtin_amdgcn_s_setprio(2);
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
    if (prio) __builtin_amdgcn_s_setprio(0);
    }   // tile with a valid key
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
    if (a.stamps && lane == 0)
      for (int k_ = 0; k_ < 5; ++k_) a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + k_] = dg[k_];
    if (a.stamps && tid == 0) a.stamps[((size_t)blockIdx.x * 4) * 8 + 7] = (unsigned long long)nt;
    // wave lifetime (entry .. end of the tile loop) and where it ran: slot 5 / 6 = s_memtime, slot 7 of waves 1..3 =
    // XCC_ID << 32 | HW_ID (tools/attn_timeline.py derives the clock and the waves resident per SIMD from these)
    unsigned long long t_end;
    uint32_t hw, xcc;
    asm volatile("s_memtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t_end), "=s"(hw), "=s"(xcc) :: "memory");
    if (a.stamps && lane == 0) {
      a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 5] = t_begin;
      a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 6] = t_end;
      if (wave) a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
  }
#undef RR_MARK
#undef RR_LOAD_TILE
#undef RR_WRITE_TILE

  // ---- epilogue: O = O^T / l ; lane writes 4 consecutive d per register group
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  {
    char* const slice = lds + wave * (32 * O_PITCH);        // K/V buffers: free since the last tile's barrier
    stage_o_rows<DT>(slice, lane & 31, h, o0, o1, inv);
    const int row0 = qblk * 128 + wave * 32;
    if (row0 < Tq)
      store_o_rows<32>(slice, lane, a.out + ((size_t)b * Tq + row0) * a.out_stride + head * 64, a.out_stride, Tq - row0);
  }
  // bound on the row sum = bound on every probability: 2^64 keeps P (bf16) times V finite in fp32; with fp16 operands P
  // itself has to stay below 65504
  constexpr float L_MAX = DT == 1 ? 6.0e4f : 1.8e19f;
  if constexpr (FIXED) return __syncthreads_or(!(l_tot < L_MAX) || need_ref) != 0;
  return false;
}

// Fixed-reference schedule with 64 query rows per wave: workgroup = 4 waves = the 256 query rows [256 qblk, 256 qblk + 256)
// of (pair, head) = grp; a wave runs TWO independent 32-query chains (sub-blocks 0 and 1) against the same K and V
// fragments.  What that buys (SQ counters of the 32-row form: waves 44 % parked, 35 % issue-stalled, MFMA pipe 30 % busy —
// a wave's tile is one serial chain K reads -> MFMA -> softmax -> MFMA): the two chains give the scheduler independent
// work to put under each other's MFMA and LDS latencies, and K/V fragments, LDS traffic and DMA pieces per query are
// halved.  ~240 VGPRs, 2 waves per SIMD.  Same arithmetic per row as attn_block<FIXED = true>; same return value.
// DIAG64: s_memtime marks per KV tile (tools/attn_timeline64.py), summed per wave into a.stamps[(block * 4 + wave) * 8 + k]:
// k = 0 next-tile DMA issue + K fragment requests, 1 QK^T issue (+ V fragment requests), 2 softmax (includes waiting for the
// QK^T results), 3 pack + P.V issue, 4 the block's PROLOGUE (once), 5 wait for the next tile's DMA + workgroup barrier,
// 6 tiles, 7 whole block.
template <int DT, bool DIAG64 = false>
__device__ __forceinline__ bool attn_block64(const int grp, const int qblk, char* const lds, const AttnArgs& a) {
  unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmk[7];
#define RR_MK(k) { if constexpr (DIAG64) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tmk[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } }
  unsigned long long t_blk0 = 0;
  if constexpr (DIAG64) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_blk0) :: "memory");
  char* const k_img = lds;
  char* const v_img = lds + 2 * TILE_BYTES;
  float* const b_all = (float*)(lds + 4 * TILE_BYTES);
  int* const f_all = (int*)(lds + 4 * TILE_BYTES + BIAS_TILES * KT * 4);
  const int Tq = a.Tq, Tk = a.Tk, kv_stride = a.kv_stride;
  const float* const key_bias = a.key_bias;
  const int b = grp / a.heads, head = grp - b * a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int qrow0 = qblk * 256 + wave * 64 + (lane & 31);        // sub-block 1: + 32

  bf16x8 qf[2][4];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const bf16_t* qp = a.q + ((size_t)((b + a.q_batch_off) / a.q_batch_div) * Tq + min(qrow0 + 32 * sb, Tq - 1)) * a.q_stride + head * 64 + 8 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) qf[sb][i] = *(const bf16x8*)(qp + 16 * i);
  }

  // K/V staging by LDS-DMA, exactly as in attn_block
  const bf16_t* kbase = uniform_ptr(a.k + (size_t)b * Tk * kv_stride + head * 64);
  const bf16_t* vbase = uniform_ptr(a.v + (size_t)b * Tk * kv_stride + head * 64);
  const uint32_t stride2 = (uint32_t)kv_stride * 2u;
  const int r0 = wave * 16 + (lane >> 3);
  const uint32_t ck0 = (uint32_t)((lane & 7) ^ (lane >> 4)) << 4;
  const uint32_t cv = (uint32_t)((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4;
  const uint32_t dst0 = __builtin_amdgcn_readfirstlane(lds_addr(k_img) + wave * 2048);
  auto load_tile = [&](const int t, const int buf) __attribute__((always_inline)) {
    const uint32_t ro0 = (uint32_t)min(t * KT + r0, Tk - 1) * stride2;
    const uint32_t ro1 = (uint32_t)min(t * KT + r0 + 8, Tk - 1) * stride2;
    const uint32_t kd = dst0 + buf * TILE_BYTES;
    glds16_so(kbase, ro0 + ck0, kd);
    glds16_so(kbase, ro1 + (ck0 ^ 64u), kd + 1024);
    glds16_so(vbase, ro0 + cv, kd + 2 * TILE_BYTES);
    glds16_so(vbase, ro1 + cv, kd + 2 * TILE_BYTES + 1024);
  };
  auto write_tile = [&](const int) __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  f32x16 o[2][2];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[sb][0][r] = 0.f; o[sb][1][r] = 0.f; }
  float l_run[2] = {0.f, 0.f};
  bool need_ref = true;
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
  const bf16x8 kone = __builtin_bit_cast(bf16x8, u32x4_{h == 0 ? (DT == 0 ? 0x3F80u : 0x3C00u) : 0u, 0u, 0u, 0u});
  bf16x8 qc[2] = {__builtin_bit_cast(bf16x8, u32x4_{0u, 0u, 0u, 0u}), __builtin_bit_cast(bf16x8, u32x4_{0u, 0u, 0u, 0u})};
  const int nt = (Tk + KT - 1) / KT;
  const bool prio = (a.tuning & 1) != 0;
  load_tile(0, 0);
  load_bias_chunk(b_all, f_all, key_bias, b, Tk, 0);
  uint32_t flag_bits = chunk_flag_bits(f_all);
  write_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);     // compiler-visible vmcnt(0): see attn_block
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (__builtin_expect(t != 0 && (t & (BIAS_TILES - 1)) == 0, 0)) {
      load_bias_chunk(b_all, f_all, key_bias, b, Tk, t / BIAS_TILES);
      flag_bits = chunk_flag_bits(f_all);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    RR_MK(0)
    if (t + 1 < nt) load_tile(t + 1, buf ^ 1);
    const char* kt_ = k_img + buf * TILE_BYTES;
    const char* vt_ = v_img + buf * TILE_BYTES;
    const float* bt_ = b_all + (t & (BIAS_TILES - 1)) * KT;
    const int tile_flags = tile_flags_of(flag_bits, t);   // bit 0: some key has a bias; bit 1: no valid key
    const bool masked = (tile_flags & 1) != 0;
    if constexpr (DIAG64) { for (int k_ = 1; k_ < 5; ++k_) tmk[k_] = tmk[0]; }
    if (!(tile_flags & 2)) {     // a tile without a valid key adds exactly 0: skipped (see attn_block)

    // all eight K fragments of the tile are requested before the first MFMA and all eight V fragments before the softmax
    // (hipcc otherwise sinks every ds_read next to its use: 4 + 4 exposed LDS latencies per tile); the scheduling fences
    // keep the requests where they are, the waits are the compiler's counted lgkmcnt
    bf16x8 kf[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kf[2 * i] = *(const bf16x8*)(kt_ + swz128(lane & 31, 2 * i + h));
      kf[2 * i + 1] = *(const bf16x8*)(kt_ + swz128(32 + (lane & 31), 2 * i + h));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIAG64) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tmk[1]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    {
    f32x16 s[2][2];     // [sub-block][key half]
    auto qk = [&]() __attribute__((always_inline)) {
      if (prio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {     // + the row's fixed reference (see attn_block)
        s[sb][0] = mfma32<DT>(kone, qc[sb], s[sb][0]);
        s[sb][1] = mfma32<DT>(kone, qc[sb], s[sb][1]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          s[sb][0] = mfma32<DT>(kf[2 * i], qf[sb][i], s[sb][0]);
          s[sb][1] = mfma32<DT>(kf[2 * i + 1], qf[sb][i], s[sb][1]);
        }
      }
      if (prio) __builtin_amdgcn_s_setprio(0);
    };
    if (masked) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b0 = *(const float4*)(bt_ + 8 * g + 4 * h);
        const float4 b1 = *(const float4*)(bt_ + 32 + 8 * g + 4 * h);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          s[sb][0][4 * g + 0] = b0.x; s[sb][0][4 * g + 1] = b0.y; s[sb][0][4 * g + 2] = b0.z; s[sb][0][4 * g + 3] = b0.w;
          s[sb][1][4 * g + 0] = b1.x; s[sb][1][4 * g + 1] = b1.y; s[sb][1][4 * g + 2] = b1.z; s[sb][1][4 * g + 3] = b1.w;
        }
      }
      qk();
    } else {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[sb][0][r] = 0.f; s[sb][1][r] = 0.f; }
      qk();
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 vf[8];       // [(kb, st)][d block]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int key0 = (ks >> 1) * 32 + 16 * (ks & 1) + 4 * h;
      const int cg = 2 * ((lane >> 4) & 1);
      vf[2 * ks] = tr_pair(vt_, key0, 0 + cg, lane);
      vf[2 * ks + 1] = tr_pair(vt_, key0, 4 + cg, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIAG64) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tmk[2]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    if (__builtin_expect(need_ref, 0)) {
      bool got = false;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        float mx = fmaxf(s[sb][0][0], s[sb][1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[sb][0][r], s[sb][1][r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx > -1e29f) {
          float c_ref;
          const uint32_t c16 = ref16<DT>(-mx, c_ref);
          qc[sb] = __builtin_bit_cast(bf16x8, u32x4_{h == 0 ? c16 : 0u, 0u, 0u, 0u});
          got = true;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[sb][0][r] += c_ref; s[sb][1][r] += c_ref; }
        }
      }
      if (got) need_ref = false;       // validity is a property of the keys: both sub-blocks and all lanes agree
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[sb][0][r] = __builtin_amdgcn_exp2f(s[sb][0][r]);
        s[sb][1][r] = __builtin_amdgcn_exp2f(s[sb][1][r]);
      }
      l_run[sb] += row_sum32(s[sb][0], s[sb][1]);
    }
    if constexpr (DIAG64) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tmk[3]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    if (prio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
          u32x4 pw;
#pragma unroll
          for (int j = 0; j < 4; ++j) pw[j] = pack2<DT>(s[sb][kb][8 * st + 2 * j], s[sb][kb][8 * st + 2 * j + 1]);
          const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
          o[sb][0] = mfma32<DT>(vf[2 * (2 * kb + st)], pf, o[sb][0]);
          o[sb][1] = mfma32<DT>(vf[2 * (2 * kb + st) + 1], pf, o[sb][1]);
        }
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
    }   // serial form
    if constexpr (DIAG64) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tmk[4]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    }   // tile with a valid key
    if (t + 1 < nt) write_tile(buf ^ 1);
    __syncthreads();
    RR_MK(5)
    if constexpr (DIAG64) {
#pragma unroll
      for (int k_ = 0; k_ < 4; ++k_) dg[k_] += tmk[k_ + 1] - tmk[k_];
      dg[5] += tmk[5] - tmk[4];           // wait for the next tile's DMA + workgroup barrier
      dg[6] += 1;
      if (t == 0) dg[4] = tmk[0] - t_blk0;     // prologue: Q loads, first K/V tile, bias chunk, barrier
    }
  }
#undef RR_MK

  constexpr float L_MAX = DT == 1 ? 6.0e4f : 1.8e19f;
  bool bad = need_ref;
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const float l_tot = l_run[sb] + __shfl_xor(l_run[sb], 32, 64);
    const float inv = 1.0f / l_tot;
    bad = bad || !(l_tot < L_MAX);
    stage_o_rows<DT>(lds + wave * (64 * O_PITCH), 32 * sb + (lane & 31), h, o[sb][0], o[sb][1], inv);   // see stage_o_rows
  }
  {
    const int row0 = qblk * 256 + wave * 64;
    if (row0 < Tq)
      store_o_rows<64>(lds + wave * (64 * O_PITCH), lane, a.out + ((size_t)b * Tq + row0) * a.out_stride + head * 64,
                       a.out_stride, Tq - row0);
  }
  if constexpr (DIAG64) {
    if (a.stamps && lane == 0) {
      unsigned long long t_end;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) :: "memory");
      dg[7] = t_end - t_blk0;
#pragma unroll
      for (int k_ = 0; k_ < 8; ++k_) a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + k_] = dg[k_];
    }
  }
  return __syncthreads_or(bad) != 0;
}

// Online form over the whole grid (DENSE bias, diagnostics, rr_set_tuning("attn_fixed_ref", 0)).
template <int DT, bool DENSE, bool DIAG = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_LDS_BYTES];
  int grp, qblk;
  if (block_map(blockIdx.x, (a.Tq + 127) >> 7, a.groups, grp, qblk)) attn_block<DT, DENSE, DIAG, false>(grp, qblk, lds, a);
}

// Fixed-reference form over the whole grid; flags[bid] = 1 where the workgroup has to be recomputed.
template <int DT, bool DIAG = false>
__global__ __launch_bounds__(256, 4) void attn_fixed_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_LDS_BYTES];
  int grp, qblk;
  const bool redo = block_map(blockIdx.x, (a.Tq + 127) >> 7, a.groups, grp, qblk) && attn_block<DT, false, DIAG, true>(grp, qblk, lds, a);
  if (threadIdx.x == 0) a.flags[blockIdx.x] = redo ? 1 : 0;
}

// Fixed-reference form, 256 query rows per workgroup (64 per wave); flags are per 256-row workgroup.
template <int DT, bool DIAG64 = false>
__global__ __launch_bounds__(256, 2) void attn_fixed64_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_LDS_BYTES];
  int grp, qblk;
  const bool redo = block_map(blockIdx.x, (a.Tq + 255) >> 8, a.groups, grp, qblk, a.rev) && attn_block64<DT, DIAG64>(grp, qblk, lds, a);
  if (threadIdx.x == 0) a.flags[blockIdx.x] = redo ? 1 : 0;
}

// Second launch of the fixed-reference schedule: workgroup i looks at flags[16 i .. 16 i + 15] and recomputes the flagged
// workgroups of the first launch in the online form.  Normally none is flagged and this is ceil(nblk / 16) workgroups
// that read 64 bytes and leave (a few microseconds); if everything is flagged (e.g. fp16 operands on data whose scores
// climb by more than 11 nats after the first tile) the recomputation is spread over as many workgroups as CUs can hold
// and costs about one online launch, not a serial tail.
constexpr int REDO_SPAN = 16;
template <int DT, int SPLIT>   // SPLIT = 128-row online workgroups per flagged workgroup of the first launch (1 or 2)
__global__ __launch_bounds__(256, 2) void attn_redo_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_LDS_BYTES];
  __shared__ int list[REDO_SPAN];
  __shared__ int count;
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  const int id = blockIdx.x * REDO_SPAN + threadIdx.x;
  if (threadIdx.x < REDO_SPAN && id < a.nblk && a.flags[id]) list[atomicAdd(&count, 1)] = id;
  __syncthreads();
  const int n = __builtin_amdgcn_readfirstlane(count);
  if (a.redo_stats && threadIdx.x == 0) {
    atomicAdd(&a.redo_stats[0], (unsigned long long)n);
    atomicAdd(&a.redo_stats[1], (unsigned long long)min(REDO_SPAN, a.nblk - (int)blockIdx.x * REDO_SPAN));
  }
  for (int j = 0; j < n; ++j) {     // workgroups are independent: the order inside the list does not matter
    int grp, qblk;
    if (!block_map(__builtin_amdgcn_readfirstlane(list[j]), (a.Tq + 128 * SPLIT - 1) / (128 * SPLIT), a.groups, grp, qblk, a.rev)) continue;
#pragma unroll
    for (int part = 0; part < SPLIT; ++part) {
      if ((qblk * SPLIT + part) * 128 < a.Tq) attn_block<DT, false, false, false>(grp, qblk * SPLIT + part, lds, a);
      __syncthreads();
    }
  }
}

// ---- packed forward (rr_forward_packed): ALL segments of a layer in one launch.  A segment = n pairs of `len` rows each
// (self-attention, Tq = Tk = len), its rows back to back from row0; blk0[s] = first workgroup of segment s in the 64-row
// form's grid (a multiple of 8: a workgroup's id modulo 8 stays its XCD inside every segment).  One launch instead of one per
// segment: no partially filled last round per segment, which is what lets the row granule shrink.
constexpr int ATTN_MAX_SEGS = 64;
struct AttnSegs {
  int nseg;
  int blk0[ATTN_MAX_SEGS + 1];
  int len[ATTN_MAX_SEGS];
  int n[ATTN_MAX_SEGS];
  long long row0[ATTN_MAX_SEGS];
};
__device__ __forceinline__ int seg_of(const AttnSegs& t, const int bid, AttnArgs& b) {      // bid wave-uniform
  int s = 0;
  while (s + 1 < t.nseg && bid >= t.blk0[s + 1]) ++s;
  const long long r0 = t.row0[s];
  b.q += r0 * b.q_stride;
  b.k += r0 * b.kv_stride;
  b.v += r0 * b.kv_stride;
  if (b.key_bias) b.key_bias += r0;
  b.out += r0 * b.out_stride;
  b.Tq = b.Tk = t.len[s];
  b.groups = t.n[s] * b.heads;
  return bid - t.blk0[s];
}
template <int DT>
__global__ __launch_bounds__(256, 2) void attn_fixed64_seg_kernel(const AttnArgs a, const AttnSegs t) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_LDS_BYTES];
  AttnArgs b = a;
  const int lbid = seg_of(t, (int)blockIdx.x, b);
  int grp, qblk;
  const bool redo = block_map(lbid, (b.Tq + 255) >> 8, b.groups, grp, qblk) && attn_block64<DT>(grp, qblk, lds, b);
  if (threadIdx.x == 0) a.flags[blockIdx.x] = redo ? 1 : 0;
}
template <int DT>
__global__ __launch_bounds__(256, 2) void attn_redo_seg_kernel(const AttnArgs a, const AttnSegs t) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_LDS_BYTES];
  __shared__ int list[REDO_SPAN];
  __shared__ int count;
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  const int id = blockIdx.x * REDO_SPAN + threadIdx.x;
  if (threadIdx.x < REDO_SPAN && id < a.nblk && a.flags[id]) list[atomicAdd(&count, 1)] = id;
  __syncthreads();
  const int n = __builtin_amdgcn_readfirstlane(count);
  if (a.redo_stats && threadIdx.x == 0) {
    atomicAdd(&a.redo_stats[0], (unsigned long long)n);
    atomicAdd(&a.redo_stats[1], (unsigned long long)min(REDO_SPAN, a.nblk - (int)blockIdx.x * REDO_SPAN));
  }
  for (int j = 0; j < n; ++j) {
    AttnArgs b = a;
    const int lbid = seg_of(t, __builtin_amdgcn_readfirstlane(list[j]), b);
    int grp, qblk;
    if (!block_map(lbid, (b.Tq + 255) >> 8, b.groups, grp, qblk)) continue;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      if ((qblk * 2 + part) * 128 < b.Tq) attn_block<DT, false, false, false>(grp, qblk * 2 + part, lds, b);
      __syncthreads();
    }
  }
}

}  // namespace

static unsigned long long* g_attn_stamps = nullptr;
// kernel argument `tuning`: bit 0 = rr_set_tuning("attn_prio"): MFMA sections at wave priority 2, softmax at 0 (+4 % attention);
// rr_set_tuning("attn_fixed_ref"): fixed-reference schedule (two launches) for grids of at least ATTN_FIXED_MIN_BLOCKS.
constexpr int ATTN_FIXED_DEFAULT = 3;   // 0 online only, 1 fixed reference (32 rows per wave), 2 fixed reference (64 rows per wave),
                                        // 3 = 2 where 256-row workgroups pad no more query rows than 128-row ones do, else 1
static int g_attn_prio_host = 1, g_attn_fixed_host = ATTN_FIXED_DEFAULT;
constexpr long ATTN_FIXED_MIN_BLOCKS = 1024;   // below this the launch, not the softmax, is what costs
extern "C" int rr_set_attn_prio(int on) { g_attn_prio_host = on != 0; return 0; }
extern "C" int rr_set_attn_fixed_ref(int v) { g_attn_fixed_host = (v < 0 || v > 3) ? ATTN_FIXED_DEFAULT : v; return 0; }   // out of range: back to the default
extern "C" int rr_get_attn_fixed_ref(void) { return g_attn_fixed_host; }
extern "C" int rr_set_attn_stamps(void* device_buf) {   // diagnostic: 4 waves x 8 uint64 per workgroup, or NULL
  g_attn_stamps = (unsigned long long*)device_buf;
  return 0;
}
static unsigned long long* g_attn_redo_stats = nullptr;
extern "C" int rr_set_attn_redo_stats(void* device_buf) {   // diagnostic: 2 x uint64 (flagged, looked at) accumulated by the redo launches, or NULL
  g_attn_redo_stats = (unsigned long long*)device_buf;
  return 0;
}

// Redo flags of the fixed-reference schedule: one word per workgroup, grow-only, one buffer per (device, stream) so that
// launches on different streams never share one.  Written in full by the first launch and read by the second.
namespace {
struct FlagBuf { int* p = nullptr; long cap = 0; };
std::mutex g_flags_mu;
std::map<std::pair<int, hipStream_t>, FlagBuf> g_flags;

hipError_t attn_flags(long nblk, hipStream_t st, int** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_flags_mu);
  FlagBuf& f = g_flags[std::make_pair(dev, st)];
  if (f.cap < nblk) {
    // growth allocates: not while the stream is being captured (rr_reserve sizes the buffer beforehand)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return hipErrorStreamCaptureUnsupported;
    // an outgrown flag buffer is NOT freed (4 bytes per workgroup; a graph captured earlier on this stream may still hold its
    // address): it stays allocated for the life of the process
    f.p = nullptr; f.cap = 0;
    const long cap = nblk + nblk / 2;
    if ((e = hipMalloc((void**)&f.p, (size_t)cap * sizeof(int))) != hipSuccess) { f.p = nullptr; return e; }
    f.cap = cap;
  }
  *out = f.p;
  return hipSuccess;
}
}  // namespace

// Pre-size the redo-flag buffer of (current device, st) for an attention launch of B sequences x heads x Tq query rows,
// so that the launch path never allocates (rr_reserve; required before capturing a forward into a hipGraph).
hipError_t rr_attention_reserve(int B, int heads, int Tq, hipStream_t st) {
  if (B <= 0 || heads <= 0 || Tq <= 0) return hipErrorInvalidValue;
  const long groups = (long)B * heads, nblk = ((groups + 7) / 8) * 8 * ((Tq + 127) / 128);
  int* p = nullptr;
  return attn_flags(nblk, st, &p);
}

hipError_t rr_launch_attention(const bf16_t* q, int q_stride, int q_batch_div, int q_batch_off, const bf16_t* k,
                               const bf16_t* v, int kv_stride, const float* key_bias, int B, int heads,
                               int Tq, int Tk, bf16_t* out, int out_stride, int dt, hipStream_t st,
                               const float* dense_bias, int dense_ld, long schedule_blocks, int fixed_mode) {
  // fixed_mode: the caller's schedule choice (a handle's "attn_fixed_ref" option), < 0 = the process-wide switch
  const int fixed_host = fixed_mode >= 0 && fixed_mode <= 3 ? fixed_mode : g_attn_fixed_host;
  // schedule_blocks > 0: the grid size that decides between the online and the fixed-reference schedule — a packed forward
  // (rr_forward_packed) launches one segment at a time and passes the PADDED call's grid, so that every pair runs the
  // schedule, hence the roundings, of the padded forward however few pairs share its length
  if (dt != 0 && dt != 1) return hipErrorInvalidValue;
  if (B <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0 || q_batch_div <= 0 || q_batch_off < 0) return hipErrorInvalidValue;
  if ((q_stride & 7) || (kv_stride & 7) || (out_stride & 7)) return hipErrorInvalidValue;     // 16-byte row chunks everywhere
  if ((long)Tk * kv_stride * 2 >= (1L << 32)) return hipErrorInvalidValue;   // LDS-DMA: 32-bit byte offsets inside one sequence
  const long groups = (long)B * heads, nqb = (Tq + 127) / 128, nblk = ((groups + 7) / 8) * 8 * nqb;
  if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
  if (dense_bias && (dense_ld < Tk || (dense_ld & 63))) return hipErrorInvalidValue;
  const dim3 grid((unsigned)nblk), block(256);
  AttnArgs a{q, q_stride, q_batch_div, q_batch_off, k, v, kv_stride, key_bias, heads, Tq, Tk, out, out_stride, (int)groups,
             dense_bias, dense_ld, nullptr, g_attn_prio_host, nullptr, (int)nblk, g_attn_redo_stats};
  const bool diag = g_attn_stamps && dt == 0 && !dense_bias;   // diagnostic timeline (tools/attn_timeline.py)
  if (diag) a.stamps = g_attn_stamps;
  if (fixed_host && !dense_bias && (schedule_blocks > 0 ? schedule_blocks : nblk) >= ATTN_FIXED_MIN_BLOCKS) {
    // 64 query rows per wave: 256-row workgroups, flags per 256-row workgroup.  Half the K/V fragment reads and DMA pieces
    // per query row (the launch is clock-limited by power: fewer LDS bytes per MFMA is what it answers to), at 2 waves per
    // SIMD; padding to 256 rows must not cost more than that saves.
    const bool rows64 = fixed_host == 2 || (fixed_host == 3 && ((Tq + 255) / 256) * 2 == (Tq + 127) / 128);
    if (rows64 && (!diag || fixed_host == 2)) {     // (stamps + attn_fixed_ref 2: the 64-row form's own timeline)
      const long nblk64 = ((groups + 7) / 8) * 8 * ((Tq + 255) / 256);
      a.nblk = (int)nblk64;
      a.rev = nblk64 >= 2048 ? rr_m_direction_next() : 0;      // large launches take part in the alternation of walking directions
      hipError_t e = attn_flags(nblk64, st, &a.flags);
      if (e != hipSuccess) return e;
      const dim3 grid64((unsigned)nblk64), rgrid((unsigned)((nblk64 + REDO_SPAN - 1) / REDO_SPAN));
      if (diag) hipLaunchKernelGGL((attn_fixed64_kernel<0, true>), grid64, block, 0, st, a);
      else if (dt == 0) hipLaunchKernelGGL((attn_fixed64_kernel<0>), grid64, block, 0, st, a);
      else hipLaunchKernelGGL((attn_fixed64_kernel<1>), grid64, block, 0, st, a);
      a.stamps = nullptr;
      if (dt == 0) hipLaunchKernelGGL((attn_redo_kernel<0, 2>), rgrid, block, 0, st, a);
      else hipLaunchKernelGGL((attn_redo_kernel<1, 2>), rgrid, block, 0, st, a);
      return hipGetLastError();
    }
    hipError_t e = attn_flags(nblk, st, &a.flags);
    if (e != hipSuccess) return e;
    const dim3 rgrid((unsigned)((nblk + REDO_SPAN - 1) / REDO_SPAN));
    if (diag) hipLaunchKernelGGL((attn_fixed_kernel<0, true>), grid, block, 0, st, a);
    else if (dt == 0) hipLaunchKernelGGL((attn_fixed_kernel<0>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_fixed_kernel<1>), grid, block, 0, st, a);
    a.stamps = nullptr;
    if (dt == 0) hipLaunchKernelGGL((attn_redo_kernel<0, 1>), rgrid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_redo_kernel<1, 1>), rgrid, block, 0, st, a);
    return hipGetLastError();
  }
  if (diag) hipLaunchKernelGGL((attn_fwd_kernel<0, false, true>), grid, block, 0, st, a);
  else if (dt == 0) { if (dense_bias) hipLaunchKernelGGL((attn_fwd_kernel<0, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((attn_fwd_kernel<0, false>), grid, block, 0, st, a); }
  else { if (dense_bias) hipLaunchKernelGGL((attn_fwd_kernel<1, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((attn_fwd_kernel<1, false>), grid, block, 0, st, a); }
  return hipGetLastError();
}

// All segments of a packed forward's layer (self-attention, fused-QKV style pointers: row r of the call at q + r * q_stride):
// one fixed-reference launch + one redo launch when `schedule_blocks` (the padded call's grid) selects that schedule, else one
// rr_launch_attention per segment (the online form of small calls).  seg_* are HOST arrays.
hipError_t rr_launch_attention_segs(const bf16_t* q, int q_stride, const bf16_t* k, const bf16_t* v, int kv_stride,
                                    const float* key_bias, int heads, int nseg, const int* seg_n, const int* seg_len,
                                    const long long* seg_row0, bf16_t* out, int out_stride, int dt, hipStream_t st,
                                    long schedule_blocks, int fixed_mode) {
  const int fixed_host = fixed_mode >= 0 && fixed_mode <= 3 ? fixed_mode : g_attn_fixed_host;
  if (dt != 0 && dt != 1) return hipErrorInvalidValue;
  if (nseg <= 0 || heads <= 0 || !seg_n || !seg_len || !seg_row0) return hipErrorInvalidValue;
  if ((q_stride & 7) || (kv_stride & 7) || (out_stride & 7)) return hipErrorInvalidValue;
  const bool one_launch = fixed_host && schedule_blocks >= ATTN_FIXED_MIN_BLOCKS && nseg <= ATTN_MAX_SEGS && !g_attn_stamps;
  if (!one_launch) {
    for (int s = 0; s < nseg; ++s) {
      const bf16_t* q0 = q + seg_row0[s] * q_stride;
      hipError_t e = rr_launch_attention(q0, q_stride, 1, 0, k + seg_row0[s] * kv_stride, v + seg_row0[s] * kv_stride, kv_stride,
                                         key_bias ? key_bias + seg_row0[s] : nullptr, seg_n[s], heads, seg_len[s], seg_len[s],
                                         out + seg_row0[s] * out_stride, out_stride, dt, st, nullptr, 0, schedule_blocks, fixed_mode);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  AttnSegs t;
  t.nseg = nseg;
  long total = 0;
  for (int s = 0; s < nseg; ++s) {
    if (seg_n[s] <= 0 || seg_len[s] <= 0 || (long)seg_len[s] * kv_stride * 2 >= (1L << 32)) return hipErrorInvalidValue;
    t.blk0[s] = (int)total;
    t.len[s] = seg_len[s];
    t.n[s] = seg_n[s];
    t.row0[s] = seg_row0[s];
    total += (((long)seg_n[s] * heads + 7) / 8) * 8 * ((seg_len[s] + 255) / 256);
    if (total > 0x7fffffffL) return hipErrorInvalidValue;
  }
  t.blk0[nseg] = (int)total;
  AttnArgs a{q, q_stride, 1, 0, k, v, kv_stride, key_bias, heads, 0, 0, out, out_stride, 0,
             nullptr, 0, nullptr, g_attn_prio_host, nullptr, (int)total, g_attn_redo_stats};
  hipError_t e = attn_flags(total, st, &a.flags);
  if (e != hipSuccess) return e;
  const dim3 grid((unsigned)total), rgrid((unsigned)((total + REDO_SPAN - 1) / REDO_SPAN)), block(256);
  if (dt == 0) hipLaunchKernelGGL((attn_fixed64_seg_kernel<0>), grid, block, 0, st, a, t);
  else hipLaunchKernelGGL((attn_fixed64_seg_kernel<1>), grid, block, 0, st, a, t);
  if (dt == 0) hipLaunchKernelGGL((attn_redo_seg_kernel<0>), rgrid, block, 0, st, a, t);
  else hipLaunchKernelGGL((attn_redo_seg_kernel<1>), rgrid, block, 0, st, a, t);
  return hipGetLastError();
}
