// bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X):  C[M,N] = epi(A[M,K] · W[N,K]^T + bias)
//
// This is the contraction behind every nn.Linear of the rerank path (QKV, attention output,
// FFN up/down, 768->128->768 bottleneck, vision MLP, mapping-network linears):
// /root/reference/src/models/rerank/rerank_model.py:374-382,418-430,461-465,557-559 and the
// HF BertLayer linears called from modeling_flmr.py:1622 / attention_fusion.py:133-144.
//
// Design (not a port of anything):
//   * both operands are K-contiguous (activations row-major, nn.Linear weight [out,in]), which is
//     exactly the per-lane 16-byte fragment of v_mfma_f32_16x16x32_bf16, so no transposes anywhere;
//   * BM x BN x 64 tile per workgroup of WM x WN waves; the big configuration is 256x256 with 8 waves
//     (2x4, 128x64 per wave = 8x4 MFMA tiles): 12 ds_read_b128 feed 32 MFMAs per 32-deep k-step, which
//     keeps the LDS at <40 % of its 256 B/clk and the L2->LDS stream at ~32 B/clk/CU;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): the LDS image is lane-linear,
//     the XOR bank swizzle is applied on the per-lane SOURCE address and again on the ds_read_b128
//     (cdna_hip_programming.md §5.4 rule 21);
//   * three kernels: gemm_kernel_s (128x128..256x256, STAGES-deep ring of whole K-tiles, one barrier per K-tile) for
//     small problems, gemm_kernel_h (256x256, ring of 8 half-tile slots, two barriers per K-tile) and gemm_kernel_hp
//     (the same loop, persistent: one workgroup per CU, the next tile's prologue requested behind the epilogue) for
//     large ones; counted s_waitcnt vmcnt(N) literals keep the youngest refills in flight across the barriers;
//   * the MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends up with 4 consecutive output columns of one output row: bias/GELU/residual epilogues
//     work on float4 and stores are 8/16-byte vectors;
//   * blockIdx -> tile map is XCD-aware (bijective remap, T1): the workgroups that share an
//     activation row-panel run on one XCD so the panel is fetched from HBM once per XCD L2.
#include "rr_common.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

// Diagnostic builds of the split residual epilogue (RR_HIPCC_EXTRA=-DRR_EPI_DIAG=n, never in a product build): bit 0 = its stores
// happen only for a value that never occurs, bit 1 = its residual loads are replaced by register constants, bit 2 = no LayerNorm
// statistics (no cross-lane sums, no partial store) — what each part costs the launch (profiles/r05_e_*); bit 3 = the `lo` half is
// loaded and stored for every second ROW of a pass only (whole cache lines skipped) (WRONG results: what a residual stream of 3 instead of 4 bytes per element
// would buy, profiles/r05_r_*); bit 4 = the statistics are computed but their partial (mean, M2) store is skipped (profiles/r05_u_*).
#ifndef RR_EPI_DIAG
#define RR_EPI_DIAG 0
#endif
// (Round 5: requesting the residual rows TWO passes ahead — two register sets of 32, the first two requests before the accumulator
// arithmetic — was built and not run: hipcc spills 69 VGPRs / 232 B of scratch at 256 registers, and a scratch access is a
// vector-memory operation that queues behind the prefetch DMA.)

namespace {

constexpr int BK = 64;
// K walk direction of an output column (all 16-bit GEMM kernels of this file, so that they keep agreeing to the bit): the
// columns of every second block of 1 024 (four 256-column slices) accumulate their K-tiles from the last to the first — see the
// serpentine note in gemm_kernel_hp, whose L2 reuse this serves.  A function of the column alone: a row's arithmetic never
// depends on M, on the tile shape or on where its tile lies.
__device__ __forceinline__ int k_walk_reversed(int n0) { return (n0 >> 10) & 1; }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt(0), i.e. waits for every global
// store / LDS-DMA in flight — between the passes of the staged epilogue that would stall on the previous pass's
// stores (and, in the persistent kernel, on the next tile's prefetch)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N < 0, "add the vmcnt literal");
}

// Wait until at most `tiles_in_flight` K-tiles (PIECES DMA instructions each) are outstanding.
template <int PIECES, int MAXT>
__device__ __forceinline__ void wait_tiles(int tiles_in_flight) {
  if constexpr (MAXT >= 2) { if (tiles_in_flight >= 2) { wait_vmcnt<2 * PIECES>(); return; } }
  if constexpr (MAXT >= 1) { if (tiles_in_flight >= 1) { wait_vmcnt<PIECES>(); return; } }
  wait_vmcnt<0>();
}

// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-erff grade) on the fast exp/rcp units:
// the libm erff costs ~10x more VALU issue slots than the whole bias/convert/store path of the epilogue.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
// tanh(x) = 1 - 2/(exp(2x)+1) on the fast exp/rcp units (abs err ~1e-7; saturates correctly at +-inf)
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
// CLIP's quick_gelu: x * sigmoid(1.702 x)
__device__ __forceinline__ float qgelu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gelu_fast(float x) { return gelu_erf_fast(x); }   // rr_common.h

// Residual taken from a not-yet-normalised row: r = LayerNorm(x) recomputed on the fly from the row statistics the
// LN kernel left behind ((x - mean) * rstd * gamma + beta, the very expression of elementwise.hip:ln_store).  This
// lets the LN kernel skip writing the fp32 normalised stream (10 -> 6 bytes per element).
struct LnResid {
  const float2* stats;   // [rows] (mean, rstd) or nullptr = `resid` already holds the residual values
  const float* gamma;
  const float* beta;
  // LayerNorm folded into the CONSUMER GEMM (north_star "fused LayerNorm+QKV"):
  //   LN(x) W^T + b = rstd * (x W'^T - mean * c) + d,  W' = 16bit(W * gamma), c_n = sum_k W'_nk, d_n = sum_k beta_k W_nk + b_n
  // producer side (EPI_BIAS_RESID_F32 only): besides the fp32 rows the epilogue writes the same values as 16-bit rows
  // `x16` (the consumer's A operand) and, per row and 128-column group, (group mean, sum of squared deviations from it);
  // ln_finalize_kernel merges the groups into (mean, rstd).  consumer side: `in_stats` (mean, rstd) per A row and
  // `csum` = c; `bias` then carries d.
  bf16_t* x16;
  int ldx;
  float2* part;          // [rows][nparts]
  int nparts;            // ceil(N / 128)
  const float2* in_stats;
  const float* csum;
  int flags;             // bit 0: touch the next pass's residual lines a pass ahead (rr_set_tuning "resid_touch"); bit 1: the lo half of the split stream is 8-bit (GemmFold::lo_bits)
  // split residual stream (GemmFold, rr_common.h): residual rows as hi + lo, output rows as x16 + lo_out
  const bf16_t* r_hi;
  const bf16_t* r_lo;
  int ld16;
  bf16_t* lo_out;
};
// Sum over aligned groups of 32 consecutive lanes, broadcast to every lane of the group, on the DPP data path (VALU
// operand modifiers: no LDS-crossbar ds_bpermute; 80 of those per pass cost the residual epilogue +24 %): a 16-lane row
// scan by row_shr 1/2/4/8 (lane 15 of a row then holds the row total), row_bcast:15 adds row 0's total into row 1 and
// row 2's into row 3, and the two group totals (lanes 31 and 63) come back through v_readlane.
__device__ __forceinline__ float dpp_f(float old, float v, int ctrl, int row_mask, bool bound) {
  return __builtin_bit_cast(float, ctrl == 0x111 ? __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)
                                   : ctrl == 0x112 ? __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true)
                                   : ctrl == 0x114 ? __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)
                                   : ctrl == 0x118 ? __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true)
                                                   : __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
}
__device__ __forceinline__ float group32_sum(float v) {
  v += dpp_f(0.f, v, 0x111, 0xf, true);      // row_shr:1 (lanes shifted in from outside the row read 0)
  v += dpp_f(0.f, v, 0x112, 0xf, true);      // row_shr:2
  v += dpp_f(0.f, v, 0x114, 0xf, true);      // row_shr:4
  v += dpp_f(0.f, v, 0x118, 0xf, true);      // row_shr:8  -> lane 15 of every row: the row's total
  v += dpp_f(0.f, v, 0x142, 0xa, false);     // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63: the group totals
  const int vi = __builtin_bit_cast(int, v);
  const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 31));
  const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 63));
  return (threadIdx.x & 32) ? hi : lo;
}
// Sum over aligned groups of 16 consecutive lanes (a DPP row), every lane of the group gets the total: four rotations.
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_rot<0x128>(v);      // row_ror:8
  v += dpp_rot<0x124>(v);      // row_ror:4
  v += dpp_rot<0x122>(v);      // row_ror:2
  v += dpp_rot<0x121>(v);      // row_ror:1
  return v;
}
// Producer side of the folded LayerNorm for one 16-byte chunk (4 consecutive columns gcol..gcol+3 of row gm) of the
// staged fp32 epilogue; lanes of an aligned group of 32 hold the 32 chunks of one 128-column group of ONE row.  Every
// lane of the wave must call this (the reductions are cross-lane); `valid` masks rows/columns outside the matrix.
template <int DT>
__device__ __forceinline__ void fold_emit(const LnResid& ln, float4 f, bool valid, int gm, int gcol, int N) {
  if (valid) store_stream(ln.x16 + (size_t)gm * ln.ldx + gcol, make_uint2(pack2<DT>(f.x, f.y), pack2<DT>(f.z, f.w)));
  const int grp = gcol >> 7;
  // columns of this group inside the matrix: 128 (reciprocal exact) except in a last, partial group
  const float rcnt = __builtin_amdgcn_rcpf((float)min(128, N - (grp << 7)));
  const float s = valid ? (f.x + f.y) + (f.z + f.w) : 0.f;
  const float m = group32_sum(s) * rcnt;
  const float a = f.x - m, b = f.y - m, c = f.z - m, d = f.w - m;
  const float q = valid ? (a * a + b * b) + (c * c + d * d) : 0.f;
  const float m2 = group32_sum(q);
  if (valid && (threadIdx.x & 31) == 0) ln.part[(size_t)gm * ln.nparts + grp] = make_float2(m, m2);
}
// Consumer side: (acc - mean * c) * rstd for 4 consecutive columns of a row with statistics st = (mean, rstd)
// ... with the bias d merged: (acc - mean c) rstd + d = acc rstd + (d - mean rstd c), two fused multiply-adds per element
// instead of three operations (|mean c| is of the size of the result here, so nothing is lost to cancellation).  Every
// kernel of this file uses this one form: they stay bit-identical to each other.
__device__ __forceinline__ f32x4 fold_apply(f32x4 acc, float2 st, float4 cs, float4 bv) {
  const float ms = -st.x * st.y;
  return f32x4{fmaf(acc[0], st.y, fmaf(ms, cs.x, bv.x)), fmaf(acc[1], st.y, fmaf(ms, cs.y, bv.y)),
               fmaf(acc[2], st.y, fmaf(ms, cs.z, bv.z)), fmaf(acc[3], st.y, fmaf(ms, cs.w, bv.w))};
}
// (Round 5: the same two multiply-adds as v_pk_fma_f32 on column pairs in the persistent ring's epilogue — bit-identical, fenced
// against the packed-f32 rule — bought nothing: 85.83 vs 85.85 ms per step, profiles/r05_q_pk_fold_ab.txt.  Removed.)
__device__ __forceinline__ float4 ln_apply(float4 x, const LnResid& ln, int gm, int gn) {
  if (!ln.stats) return x;
  float2 st = ln.stats[gm];
  const float4 g = *(const float4*)(ln.gamma + gn), b = *(const float4*)(ln.beta + gn);
  // gfx950 hazard (found with tools/slp_hazard_probe.py, ISA and measurements in DESIGN.md "Numerics"): when hipcc forms
  // packed fp32 code here (-fslp-vectorize) it emits `s_waitcnt vmcnt(N)` DIRECTLY followed by `v_pk_add_f32 v[x:x+1], ...`
  // on the registers the load just returned, and on lanes 48-63 the LOW register of the pair is then intermittently read
  // stale (16 rows x 1 column of a tile, different tiles every launch, under two workgroups per CU).  Any instruction
  // between the wait and the first packed consumer cures it (0 of 4 launches bad against 4 of 4; an empty asm statement is
  // enough because hipcc pads its boundary with `s_nop 0`).  So: make the loaded values opaque here — the compiler's
  // wait lands before this statement — and spend two wait states before anything consumes them.  The build also keeps
  // -fno-slp-vectorize (no compiler-formed v_pk_*_f32 anywhere) as the second line of defence.
  asm volatile("s_nop 1" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w), "+v"(st.x), "+v"(st.y));
  return make_float4((x.x - st.x) * st.y * g.x + b.x, (x.y - st.x) * st.y * g.y + b.y,
                     (x.z - st.x) * st.y * g.z + b.z, (x.w - st.x) * st.y * g.w + b.w);
}

// ---- diagnostic cycle stamps (tools/bench_gemm.py --stamps): block entry / first tile ready / main loop done /
// epilogue done, written by lane 0 of wave 0 into a buffer no kernel reads.  nullptr in every product launch.
// De-synchronise the CUs: all workgroups of the first dispatch wave would otherwise reach their store epilogue at
// the same instant and fight for HBM write bandwidth (measured: 128 KiB/CU epilogues run at the chip-wide
// HBM write rate while the main loops leave HBM idle).  Workgroup b of the first `first_wave` ones sleeps
// ((b>>3)&7) * unit cycles once; later workgroups inherit the skew because they start when a CU frees up.
__device__ __forceinline__ void start_stagger(int first_wave, int unit_sleeps) {
  if ((int)blockIdx.x < first_wave) {
    const int n = ((blockIdx.x >> 3) & 7) * unit_sleeps;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);     // 127 * 64 cycles each
  }
}

__device__ __forceinline__ void stamp(unsigned long long* stamps, int slot) {
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
}

// Variant S ("simple"): 16x16x32 MFMA, tile barrier at the top of each K-tile, fragment reads scheduled by the
// compiler inside the tile.
template <int BM, int BN, int WM, int WN, int STAGES, int EPI, bool LDS_EPI, int DT>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel_s(const bf16_t* __restrict__ A, int lda,
                                                             const bf16_t* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, int ldr,
                                                             void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                             int tiles_n, int nwg, unsigned long long* stamps,
                                                             int stagger_unit, LnResid ln) {
  if (stagger_unit > 0 && stagger_unit < 60) start_stagger(256, stagger_unit);
  stamp(stamps, 0);
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16;            // 16x16 MFMA tiles per wave
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW;
  constexpr int PIECES = PA + PW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into whole pieces per wave");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // per-piece source = scalar base (A or W, advanced by k0) + a 32-bit per-lane byte offset that lives in ONE VGPR for
  // the whole kernel: nothing rewrites an address register behind an LDS-DMA that may still be waiting to issue
  // The 64-bit part of the address (tile origin) is a scalar; the 32-bit per-lane offset only spans the tile's own rows
  // (< 256 rows x row pitch), so operands of any size are addressed correctly (no 4 GiB limit on A or W).
  uint32_t a_off[PA], w_off[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave * PA + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    a_off[i] = (uint32_t)(((size_t)min(r, M - 1 - m0) * lda + c * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = (wave * PW + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w_off[i] = (uint32_t)(((size_t)min(r, N - 1 - n0) * ldw + c * 8) * 2);
  }
  const bf16_t* const a_tile = A + (size_t)m0 * lda;
  const bf16_t* const w_tile = W + (size_t)n0 * ldw;
  const uint32_t lds_base = lds_addr(lds);
  auto stage = [&](int buf, int k0) {
    const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + wave * PA * 1024);
    const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + A_BYTES + wave * PW * 1024);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16_so(a_tile + k0, a_off[i], a_dst + i * 1024);
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16_so(w_tile + k0, w_off[i], w_dst + i * 1024);
  };

  f32x4 acc[NT][MT];   // lane holds m = mt*16 + (lane&15), n = nt*16 + (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = Kd / BK;
  const int krev = k_walk_reversed(n0);
  auto kof = [&](int idx) { return (krev ? nk - 1 - idx : idx) * BK; };     // element offset of the idx-th K-tile of the walk
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, kof(s));

  unsigned long long t_wait = 0, t_bar = 0;   // diagnostic accumulators (only when stamps != nullptr)
  for (int kt = 0; kt < nk; ++kt) {
    unsigned long long tA = 0, tB = 0;
    if (stamps) tA = __builtin_amdgcn_s_memtime();
    wait_tiles<PIECES, STAGES - 2>(min(nk - 1, kt + STAGES - 2) - kt);
    if (stamps) tB = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();          // everyone's pieces of tile kt landed; compute(kt-1) is done
    if (stamps) { const unsigned long long tC = __builtin_amdgcn_s_memtime(); t_wait += tB - tA; t_bar += tC - tB; }
    if (kt == 0) stamp(stamps, 1);
    if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES, kof(kt + STAGES - 1));
    const char* a_t = lds + (kt % STAGES) * STAGE_BYTES;
    const char* w_t = a_t + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MT], wf[NT];
      const int c = ks * 4 + (lane >> 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t] = *(const bf16x8*)(w_t + swz128(wn * TN + t * 16 + (lane & 15), c));
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = *(const bf16x8*)(a_t + swz128(wm * TM + t * 16 + (lane & 15), c));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = mfma16<DT>(wf[nt], af[mt], acc[nt][mt]);
    }
  }
  stamp(stamps, 2);
  if (stamps && threadIdx.x == 0) { stamps[(size_t)blockIdx.x * 8 + 4] = t_wait; stamps[(size_t)blockIdx.x * 8 + 5] = t_bar; }

  constexpr bool F32_OUT = (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RESID_F32);
  if constexpr (LDS_EPI) {
    // ---- epilogue through LDS: the direct form issues 32 scattered 8-byte stores per lane (16 rows x 32 B per
    // wave-instruction) and measured store-ISSUE-bound at ~9 B/clk/CU.  Here each wave group drops its finished
    // values (bias/activation applied, final dtype) into a row-major staging image, then all waves stream it out
    // as 16 B per lane along rows: every wave-instruction covers whole 512-byte (bf16) / 1-KiB (f32) row segments,
    // and the fp32 residual is read with the same coalesced pattern and added on the way out.
    constexpr int ES = F32_OUT ? 4 : 2;
    constexpr int PITCH = BN * ES + 16;                       // +16 B: rows start on different banks
    constexpr int CPR = BN * ES / 16;                         // 16-byte chunks per row
    constexpr int NTHR = NW * 64;
    // bf16 images of all WM row groups fit LDS at once (256 x 528 B); f32 images go one row group at a time
    constexpr int NPASS = F32_OUT ? WM : 1;
    constexpr int ROWS = F32_OUT ? TM : BM;
    // (1) every wave finishes its own values first (bias + activation, final dtype) so that the GELU/tanh VALU work
    //     of all 8 waves runs concurrently rather than one row group at a time
    uint2 pk[F32_OUT ? 1 : NT][F32_OUT ? 1 : MT];
    float2 fst[MT];          // folded LayerNorm, consumer side: (mean, rstd) of this lane's rows
    if (ln.in_stats) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fst[mt] = ln.in_stats[min(m0 + wm * TM + mt * 16 + (lane & 15), M - 1)];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int gn = n0 + wn * TN + nt * 16 + (lane >> 4) * 4;
      const float4 bv = (bias && gn < N) ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ln.in_stats && gn < N) cs = *(const float4*)(ln.csum + gn);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 ab = ln.in_stats ? fold_apply(acc[nt][mt], fst[mt], cs, bv)
                                     : f32x4{acc[nt][mt][0] + bv.x, acc[nt][mt][1] + bv.y, acc[nt][mt][2] + bv.z, acc[nt][mt][3] + bv.w};
        float v0 = ab[0], v1 = ab[1], v2 = ab[2], v3 = ab[3];
        if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
        if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
        if (EPI == EPI_BIAS_QGELU_BF16) { v0 = qgelu_fast(v0); v1 = qgelu_fast(v1); v2 = qgelu_fast(v2); v3 = qgelu_fast(v3); }
        if constexpr (F32_OUT) acc[nt][mt] = f32x4{v0, v1, v2, v3};
        else pk[nt][mt] = make_uint2(pack2<DT>(v0, v1), pack2<DT>(v2, v3));
      }
    }
    __builtin_amdgcn_s_barrier();                             // all MFMA-phase LDS reads are complete
    for (int pass = 0; pass < NPASS; ++pass) {
      if (!F32_OUT || wm == pass) {
        const int rbase = F32_OUT ? 0 : wm * TM;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int cn = wn * TN + nt * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            char* dst = lds + (rbase + mt * 16 + (lane & 15)) * PITCH + cn * ES;
            if constexpr (F32_OUT) *(f32x4*)dst = acc[nt][mt];
            else *(uint2*)dst = pk[nt][mt];
          }
        }
      }
      lds_barrier();
      const int row_base = m0 + (F32_OUT ? pass * TM : 0);
      static_assert((ROWS * CPR) % NTHR == 0, "every thread runs the same number of stream-out steps (cross-lane sums below)");
#pragma unroll 4
      for (int i = tid; i < ROWS * CPR; i += NTHR) {
        const int r = i / CPR, c = i - r * CPR;
        const int gm = row_base + r, gcol = n0 + c * (16 / ES);
        const bool ok = gm < M && gcol < N;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          uint4 v = *(const uint4*)(lds + r * PITCH + c * 16);
          if (EPI == EPI_BIAS_RESID_F32) {
            const float4 rv = ln_apply(*(const float4*)(resid + (size_t)gm * ldr + gcol), ln, gm, gcol);
            f = __builtin_bit_cast(float4, v);
            f.x += rv.x; f.y += rv.y; f.z += rv.z; f.w += rv.w;
            v = __builtin_bit_cast(uint4, f);
          }
          *(uint4*)((char*)Cv + ((size_t)gm * ldc + gcol) * ES) = v;
        }
        // folded LayerNorm, producer side (an aligned group of 32 lanes = the 32 chunks of one 128-column group of a row)
        if constexpr (EPI == EPI_BIAS_RESID_F32 && CPR % 32 == 0) { if (ln.x16) fold_emit<DT>(ln, f, ok, gm, gcol, N); }
      }
      if (pass + 1 < NPASS) lds_barrier();
    }
  } else {
  float2 fst[MT];            // folded LayerNorm, consumer side: (mean, rstd) of this lane's rows
  if (ln.in_stats) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fst[mt] = ln.in_stats[min(m0 + wm * TM + mt * 16 + (lane & 15), M - 1)];
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int gn = n0 + wn * TN + nt * 16 + (lane >> 4) * 4;
    if (gn >= N) continue;
    float4 bv = bias ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 cs = ln.in_stats ? *(const float4*)(ln.csum + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gm = m0 + wm * TM + mt * 16 + (lane & 15);
      if (gm >= M) continue;
      const f32x4 ab = ln.in_stats ? fold_apply(acc[nt][mt], fst[mt], cs, bv)
                                   : f32x4{acc[nt][mt][0] + bv.x, acc[nt][mt][1] + bv.y, acc[nt][mt][2] + bv.z, acc[nt][mt][3] + bv.w};
      float v0 = ab[0], v1 = ab[1], v2 = ab[2], v3 = ab[3];
      if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
      if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
      if (EPI == EPI_BIAS_QGELU_BF16) { v0 = qgelu_fast(v0); v1 = qgelu_fast(v1); v2 = qgelu_fast(v2); v3 = qgelu_fast(v3); }
      if (EPI == EPI_BIAS_RESID_F32) {
        const float4 rv = ln_apply(*(const float4*)(resid + (size_t)gm * ldr + gn), ln, gm, gn);
        v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
      }
      if (F32_OUT) {
        *(float4*)((float*)Cv + (size_t)gm * ldc + gn) = make_float4(v0, v1, v2, v3);
      } else {
        *(uint2*)((bf16_t*)Cv + (size_t)gm * ldc + gn) = make_uint2(pack2<DT>(v0, v1), pack2<DT>(v2, v3));
      }
    }
  }
  }
  stamp(stamps, 3);
}

// Variant H ("half-tile ring"): 256x256x64 K-tiles, 8 waves (wr = wave>>2, wc = wave&3).  The LDS ring holds
// 8 half-tile slots of 16 KiB: {A rows 0-127, B rows 0-127, B rows 128-255, A rows 128-255} x 2 tile parities, and a
// slot is refilled by LDS-DMA as soon as its fragments have been copied to registers, so ~5 half-tiles (80 KiB)
// stay in flight instead of one drained 64-KiB tile (tools/dma_probe.hip: 66 vs 53 GB/s per CU).  Every K-tile
// is 4 phases, one output quadrant (64x32 per wave, 16 MFMAs) each: (A0,B0) (A0,B1) (A1,B1) (A1,B0); the
// fragments a phase needs are read from LDS during the PREVIOUS phase, behind that phase's MFMAs.
//   phase:  [ds_read fragments for the next 8-MFMA block] [DMA of a half-tile whose slot is free] [8 MFMAs] x 2
//   two sync points per K-tile (X after p1, Y after p3): [s_waitcnt vmcnt(N)] [lgkmcnt(0)] [s_barrier]
// RAW: every wave waits for its own DMA pieces before the barrier that precedes their first ds_read.
// WAR: a slot is refilled only after the barrier that follows its last read.  N counts exactly the DMA instructions
// issued after the needed half-tile (2 per half-tile per wave), so the wait is exact in the tail as well.
template <int EPI, bool LDS_EPI, int DT, bool DIAG = false>
__global__ __launch_bounds__(512) void gemm_kernel_h(const bf16_t* __restrict__ A, int lda,
                                                     const bf16_t* __restrict__ W, int ldw,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ resid, int ldr,
                                                     void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                     int tiles_n, int nwg, unsigned long long* stamps, LnResid ln,
                                                     int stagger_unit) {
  if (stagger_unit > 0 && stagger_unit < 60) start_stagger(256, stagger_unit);
  if (stagger_unit == 63) lda = 0;                          // diagnostic: every A row is row 0 (cache-resident A; wrong results)
  if (stagger_unit == 62) { lda = 0; ldw = 0; }             // diagnostic: A and W cache-resident
  stamp(stamps, 0);
  constexpr int BM = 256, BN = 256, HALF = 128 * 128;       // half-tile = 128 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- DMA sources: half-tile j in {0:A0, 1:B0, 2:B1, 3:A1}; this wave moves pieces 2*wave, 2*wave+1 (8 rows each).
  // 32-bit byte offsets from the scalar origin of THIS tile's rows (a_tile / w_tile): they span at most 256 row
  // pitches, so A and W may be of any size.
  uint32_t so_a0[2], so_a1[2], so_b0[2], so_b1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + (lane >> 3);          // row inside the half-tile
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    so_a0[i] = (uint32_t)(((size_t)min(r, M - 1 - m0) * lda + c * 8) * 2);
    so_a1[i] = (uint32_t)(((size_t)min(128 + r, M - 1 - m0) * lda + c * 8) * 2);
    so_b0[i] = (uint32_t)(((size_t)min(r, N - 1 - n0) * ldw + c * 8) * 2);
    so_b1[i] = (uint32_t)(((size_t)min(128 + r, N - 1 - n0) * ldw + c * 8) * 2);
  }
  const bf16_t* const a_tile = A + (size_t)m0 * lda;
  const bf16_t* const w_tile = W + (size_t)n0 * ldw;
  const uint32_t lds_base = lds_addr(lds);
  const int nk = Kd / BK, H = 4 * nk;
  const int krev = k_walk_reversed(n0);
  // half-tile (t, J): slot = parity (t&1) * 4 + J; J is a compile-time constant at every call site
#define RR_DMA(t_, J)                                                                                             \
  {                                                                                                               \
    const uint32_t dst_ = __builtin_amdgcn_readfirstlane(lds_base + (((t_) & 1) * 4 + (J)) * HALF + wave * 2048); \
    const int tk_ = krev ? nk - 1 - (t_) : (t_);             /* K-tile t of the walk (k_walk_reversed) */                    \
    const void* sb_ = ((J) == 0 || (J) == 3) ? (const void*)(a_tile + (size_t)tk_ * BK) : (const void*)(w_tile + (size_t)tk_ * BK); \
    const uint32_t* so_ = (J) == 0 ? so_a0 : (J) == 3 ? so_a1 : (J) == 1 ? so_b0 : so_b1;                         \
    glds16_so(sb_, so_[0], dst_);                                                                                 \
    glds16_so(sb_, so_[1], dst_ + 1024);                                                                          \
  }
  // wait until half-tile h_need (and everything older) has landed; h_last = newest half-tile issued so far
  auto wait_half = [&](int h_need, int h_last) {
    if (h_need >= H) return;
    const int after = h_last - h_need;                        // half-tiles issued after the needed one
    if (after >= 3) wait_vmcnt<6>();
    else if (after == 2) wait_vmcnt<4>();
    else if (after == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
  };

  // ---- fragment addresses inside a half-tile slot (swizzle term is the same for every 16-row block); the two
  // 32-deep k-steps of a fragment differ by chunk ^ 4 = byte offset ^ 64
  const int a_off = swz128(wr * 64 + (lane & 15), lane >> 4);
  const int b_off = swz128(wc * 32 + (lane & 15), lane >> 4);
  auto read_a = [&](const char* slot, int ks, bf16x8 (&f)[4]) {
    const char* b = slot + (ks ? (a_off ^ 64) : a_off);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) f[mt] = *(const bf16x8*)(b + mt * 2048);
  };
  auto read_b = [&](const char* slot, int ks, bf16x8 (&f)[2]) {
    const char* b = slot + (ks ? (b_off ^ 64) : b_off);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) f[nt] = *(const bf16x8*)(b + nt * 2048);
  };

  f32x4 acc[4][2][4];   // [quadrant 2*hA+hB][nt][mt]; lane: m = mt*16 + (lane&15), n = nt*16 + (lane>>4)*4 + reg
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Register fragments: A of the current row half (both k-steps), B0 (kept p0..p3) and B1 (p1..p2).  Fragments for a
  // k-step are fetched one 8-MFMA block ahead of their use, so nothing is double-buffered: 64 fragment + 128
  // accumulator registers.
  bf16x8 AF0[4], AF1[4], B0K0[2], B0K1[2], B1K0[2], B1K1[2];

#define RR_BLK(Q, AF, BF)                                                                          \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)  \
      acc[Q][nt][mt] = mfma16<DT>(BF[nt], AF[mt], acc[Q][nt][mt]);
#define RR_SBAR() __builtin_amdgcn_sched_barrier(0)
  // wave priority falls with progress inside a barrier interval: of the two waves that share a SIMD the one that is
  // BEHIND wins the MFMA pipe, so they advance together instead of the older wave racing ahead to idle at the barrier
#define RR_PRIO(p) __builtin_amdgcn_s_setprio(p);

  // ---- prologue: half-tiles 0..6 in flight (g = 4*tile + {A0:0, B0:1, B1:2, A1:3}); first fragments of tile 0
  RR_DMA(0, 0) RR_DMA(0, 1) RR_DMA(0, 2) RR_DMA(0, 3)
  if (nk > 1) { RR_DMA(1, 0) RR_DMA(1, 1) RR_DMA(1, 2) }
  wait_half(3, min(H - 1, 6));                              // all of tile 0 landed (my pieces)
  __builtin_amdgcn_s_barrier();
  stamp(stamps, 1);
  read_a(lds + 0 * HALF, 0, AF0);
  read_b(lds + 1 * HALF, 0, B0K0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(AF0[0]), "+v"(AF0[1]), "+v"(AF0[2]), "+v"(AF0[3]), "+v"(B0K0[0]), "+v"(B0K0[1]));

  // Diagnostic build (DIAG): per-wave s_memtime marks inside every phase, summed over the loop and written to
  // stamps[block][8 + wave*8 + k]; the marks are read only after the phase's own lgkmcnt(0), so they add no wait.
  unsigned long long dg[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tmk[9];
#define RR_MARK(k) { if constexpr (DIAG) { RR_SBAR(); asm volatile("s_memtime %0" : "=s"(tmk[k]) :: "memory"); RR_SBAR(); } }
#define RR_ACC(base, n)                                                                       \
  {                                                                                           \
    if constexpr (DIAG) {                                                                     \
      _Pragma("unroll") for (int k_ = 0; k_ < (n); ++k_) dg[(base) + k_] += tmk[k_ + 1] - tmk[k_]; \
    }                                                                                         \
  }
  // sync point with marks m0 (before), m0+1 (after the vmcnt wait), m0+2 (after lgkmcnt(0)), m0+3 (after the barrier).
  // In the steady state the wait is a literal (no scalar branch cascade in the loop body).
#define RR_SYNC(STEADY, NLIT, g_need, g_last, m0)        \
  {                                                      \
    RR_SBAR();                                           \
    RR_MARK(m0)                                          \
    if (STEADY) wait_vmcnt<NLIT>();                      \
    else wait_half(g_need, g_last);                      \
    RR_MARK((m0) + 1)                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    RR_MARK((m0) + 2)                                    \
    __builtin_amdgcn_s_barrier();                        \
    RR_SBAR();                                           \
    RR_MARK((m0) + 3)                                    \
  }

  // Two barriers per K-tile.  X (end of p1): A0/B0 of the next tile have landed and every wave is done with this
  // tile's A0, B0, B1 slots; Y (end of p3): B1/A1 of the next tile have landed and every wave is done with this tile's
  // A1 slot.  p0 and p2 end without a barrier: what they hand to p1 / p3 was covered by the previous Y / X.  (Four
  // barriers, one per phase, cost 17 % more main-loop cycles.)  Refills: p0 A1(t+1); p2 A0(t+2), B0(t+2); p3 B1(t+2) —
  // the g order the counted waits rely on.  Measured and NOT better: one barrier per tile (the refill window shrinks
  // to one tile and the drained DMA sets the pace), an LDS arrive/poll counter instead of s_barrier (s_barrier itself
  // is ~36 cycles, tools/barrier_probe.hip), one 1-KiB piece per 8-MFMA block instead of pairs (a piece costs its wave
  // ~60 cycles wherever it sits), all pieces moved by the younger wave of each SIMD pair under EXEC masking (-10 %).
  // RR_TILE(1) is the steady state (t <= nk-3): every refill exists and the vmcnt waits are literals, so the body is
  // straight-line code; RR_TILE(0) handles the last two tiles with the general guards.
#define RR_TILE(STEADY)                                                                                    \
  {                                                                                                        \
    const char* sl = lds + (t & 1) * 4 * HALF;               /* this tile's slots: +0 A0, +1 B0, +2 B1, +3 A1 */ \
    const char* sn = lds + ((t + 1) & 1) * 4 * HALF;         /* next tile's */                              \
    const bool d1 = ((STEADY) || t + 1 < nk) && !no_dma, d2 = ((STEADY) || t + 2 < nk) && !no_dma;          \
    /* ---- p0: quadrant (A0, B0) */                                                                        \
    RR_MARK(0)                                                                                             \
    read_a(sl + 0 * HALF, 1, AF1);                                                                         \
    read_b(sl + 1 * HALF, 1, B0K1);                                                                        \
    RR_SBAR();                                                                                             \
    RR_PRIO(3)                                                                                             \
    RR_BLK(0, AF0, B0K0)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_MARK(1)                                                                                             \
    read_b(sl + 2 * HALF, 0, B1K0);                                                                        \
    if (d1) RR_DMA(t + 1, 3)                                 /* A1 of the next tile (slot free since Y(t-1)) */ \
    RR_SBAR();                                                                                             \
    RR_MARK(2)                                                                                             \
    RR_PRIO(2)                                                                                             \
    RR_BLK(0, AF1, B0K1)                                                                                   \
    RR_SBAR();                                                                                             \
    /* ---- p1: quadrant (A0, B1) */                                                                        \
    RR_MARK(3)                                                                                             \
    read_b(sl + 2 * HALF, 1, B1K1);                                                                        \
    RR_SBAR();                                                                                             \
    RR_PRIO(1)                                                                                             \
    RR_BLK(1, AF0, B1K0)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_MARK(4)                                                                                             \
    read_a(sl + 3 * HALF, 0, AF0);                                                                         \
    RR_SBAR();                                                                                             \
    RR_PRIO(0)                                                                                             \
    RR_BLK(1, AF1, B1K1)                                                                                   \
    RR_SYNC(STEADY, 4, 4 * (t + 1) + 1, min(H - 1, 4 * (t + 1) + 3), 5)   /* X: A0(t+1), B0(t+1) landed */    \
    RR_ACC(0, 8)                                                                                           \
    /* ---- p2: quadrant (A1, B1) */                                                                        \
    RR_MARK(0)                                                                                             \
    read_a(sl + 3 * HALF, 1, AF1);                                                                         \
    if (d2) RR_DMA(t + 2, 0)                                 /* slots A0, B0 (free since X) */               \
    RR_SBAR();                                                                                             \
    RR_PRIO(3)                                                                                             \
    RR_BLK(3, AF0, B1K0)                                                                                   \
    RR_SBAR();                                                                                             \
    if (d2) RR_DMA(t + 2, 1)                                                                               \
    RR_SBAR();                                                                                             \
    RR_PRIO(2)                                                                                             \
    RR_BLK(3, AF1, B1K1)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_MARK(1)                                                                                             \
    /* ---- p3: quadrant (A1, B0) */                                                                        \
    RR_PRIO(1)                                                                                             \
    RR_BLK(2, AF0, B0K0)                                                                                   \
    RR_SBAR();                                                                                             \
    if ((STEADY) || t + 1 < nk) {                                                                          \
      read_a(sn + 0 * HALF, 0, AF0);                                                                       \
      read_b(sn + 1 * HALF, 0, B0K0);                                                                      \
    }                                                                                                      \
    if (d2) RR_DMA(t + 2, 2)                                 /* slot B1 (free since X) */                    \
    RR_SBAR();                                                                                             \
    RR_PRIO(0)                                                                                             \
    RR_BLK(2, AF1, B0K1)                                                                                   \
    RR_SYNC(STEADY, 6, 4 * (t + 1) + 3, min(H - 1, 4 * (t + 2) + 2), 2)   /* Y: B1(t+1), A1(t+1) landed */    \
    /* The next tile's first fragments are complete here (lgkmcnt(0) above).  Tell the compiler: otherwise it    \
       treats them as pending across the back edge and puts lgkmcnt(0) behind the six ds_reads that open p0. */  \
    asm volatile("" : "+v"(AF0[0]), "+v"(AF0[1]), "+v"(AF0[2]), "+v"(AF0[3]), "+v"(B0K0[0]), "+v"(B0K0[1])); \
    RR_ACC(8, 5)                                                                                           \
  }
  const bool no_dma = DIAG && stagger_unit == 61;            // diagnostic: main loop without refills (wrong results)
  int t = 0;
  for (; t < nk - 2; ++t) RR_TILE(1)
  for (; t < nk; ++t) RR_TILE(0)
#undef RR_TILE
  if constexpr (DIAG) {
    if (stamps && lane == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned long long* o = stamps + (size_t)gridDim.x * 8 + ((size_t)blockIdx.x * 8 + wave) * 16;
      for (int k = 0; k < 13; ++k) o[k] = dg[k];
    }
  }
#undef RR_SYNC
#undef RR_MARK
#undef RR_ACC
#undef RR_DMA
#undef RR_BLK
#undef RR_SBAR
  wait_vmcnt<0>();   // nothing is in flight any more (every issued half-tile was waited for); explicit before LDS reuse
  stamp(stamps, 2);

  // ---- epilogue (quadrant q = 2*hA + hB): rows hA*128 + wr*64 + mt*16 + (lane&15), cols hB*128 + wc*32 + nt*16 + (lane>>4)*4
  constexpr bool F32_OUT = (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RESID_F32);
  // folded LayerNorm, consumer side (both epilogue forms): (mean, rstd) of the 8 rows this lane owns
  float2 fst[2][4];
  if (ln.in_stats) {
#pragma unroll
    for (int hA = 0; hA < 2; ++hA)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        fst[hA][mt] = ln.in_stats[min(m0 + hA * 128 + wr * 64 + mt * 16 + (lane & 15), M - 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int gn = n0 + (q & 1) * 128 + wc * 32 + nt * 16 + (lane >> 4) * 4;
        const float4 cs = gn < N ? *(const float4*)(ln.csum + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 bf = (bias && gn < N) ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);   // merged here, not added again below
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[q][nt][mt] = fold_apply(acc[q][nt][mt], fst[q >> 1][mt], cs, bf);
      }
  }
  const bool bias_done = ln.in_stats != nullptr;
  if constexpr (LDS_EPI) {
    constexpr int ES = F32_OUT ? 4 : 2;
    constexpr int PITCH = BN * ES + 16;
    constexpr int CPR = BN * ES / 16;
    constexpr int NPASS = F32_OUT ? 2 : 1;                   // f32: one 128-row half (hA) per pass
    constexpr int ROWS = F32_OUT ? 128 : 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hB = q & 1;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int gn = n0 + hB * 128 + wc * 32 + nt * 16 + (lane >> 4) * 4;
        const float4 bv = (bias && gn < N && !bias_done) ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float v0 = acc[q][nt][mt][0] + bv.x, v1 = acc[q][nt][mt][1] + bv.y, v2 = acc[q][nt][mt][2] + bv.z,
                v3 = acc[q][nt][mt][3] + bv.w;
          if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
          if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
          if (EPI == EPI_BIAS_QGELU_BF16) { v0 = qgelu_fast(v0); v1 = qgelu_fast(v1); v2 = qgelu_fast(v2); v3 = qgelu_fast(v3); }
          acc[q][nt][mt] = f32x4{v0, v1, v2, v3};
        }
      }
    }
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hA = q >> 1, hB = q & 1;
        if (F32_OUT && hA != pass) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int cn = hB * 128 + wc * 32 + nt * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int r = (F32_OUT ? 0 : hA * 128) + wr * 64 + mt * 16 + (lane & 15);
            char* dst = lds + r * PITCH + cn * ES;
            if constexpr (F32_OUT) *(f32x4*)dst = acc[q][nt][mt];
            else *(uint2*)dst = make_uint2(pack2<DT>(acc[q][nt][mt][0], acc[q][nt][mt][1]),
                                           pack2<DT>(acc[q][nt][mt][2], acc[q][nt][mt][3]));
          }
        }
      }
      lds_barrier();
      const int row_base = m0 + (F32_OUT ? pass * 128 : 0);
      // stream the staged rows out, UNR 16-byte chunks per thread in flight at a time: the residual loads of a whole
      // batch are issued before the first add/store so that ~64-128 KiB per CU are outstanding (the 4-deep form ran
      // at ~8 B/clk/CU, below the ~12 B/clk/CU a pure streaming kernel reaches)
      constexpr int UNR = (EPI == EPI_BIAS_RESID_F32) ? 16 : 8;   // f32 residual pass: the whole 128-row pass in one batch
      // 512 threads cover 512/CPR whole rows per step, so a thread keeps ONE column chunk for the whole pass (its
      // gamma/beta are loaded once) and a wave keeps one row per step (its LayerNorm statistics are a scalar load).
      static_assert(512 % CPR == 0, "a wave must not straddle rows");
      const int my_col = n0 + (tid % CPR) * (16 / ES);
      float4 lg = make_float4(1.f, 1.f, 1.f, 1.f), lb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == EPI_BIAS_RESID_F32 && ln.stats && my_col < N) {
        lg = *(const float4*)(ln.gamma + my_col);
        lb = *(const float4*)(ln.beta + my_col);
      }
      for (int i0 = tid; i0 < ROWS * CPR; i0 += 512 * UNR) {
        float4 rv[UNR];
        if (EPI == EPI_BIAS_RESID_F32) {
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * 512, r = i / CPR, c = i - r * CPR;
            const int gm = row_base + r, gcol = n0 + c * (16 / ES);
            const bool ok = i < ROWS * CPR && gm < M && gcol < N;
            float4 x = ok ? *(const float4*)(resid + (size_t)gm * ldr + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ln.stats) {
              const int gm_s = __builtin_amdgcn_readfirstlane(min(gm, M - 1));     // wave-uniform row
              const float2 st2 = ln.stats[gm_s];
              x = make_float4((x.x - st2.x) * st2.y * lg.x + lb.x, (x.y - st2.x) * st2.y * lg.y + lb.y,
                              (x.z - st2.x) * st2.y * lg.z + lb.z, (x.w - st2.x) * st2.y * lg.w + lb.w);
            }
            rv[u] = x;
          }
        }
        static_assert((ROWS * CPR) % (512 * UNR) == 0, "every thread runs every step (cross-lane sums below)");
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int i = i0 + u * 512, r = i / CPR, c = i - r * CPR;
          const int gm = row_base + r, gcol = n0 + c * (16 / ES);
          const bool ok = i < ROWS * CPR && gm < M && gcol < N;
          float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) {
            uint4 v = *(const uint4*)(lds + r * PITCH + c * 16);
            if (EPI == EPI_BIAS_RESID_F32) {
              f = __builtin_bit_cast(float4, v);
              f.x += rv[u].x; f.y += rv[u].y; f.z += rv[u].z; f.w += rv[u].w;
              v = __builtin_bit_cast(uint4, f);
            }
            *(uint4*)((char*)Cv + ((size_t)gm * ldc + gcol) * ES) = v;
          }
          if (EPI == EPI_BIAS_RESID_F32) { if (ln.x16) fold_emit<DT>(ln, f, ok, gm, gcol, N); }   // folded LayerNorm, producer side
        }
      }
      if (pass + 1 < NPASS) lds_barrier();
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hA = q >> 1, hB = q & 1;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int gn = n0 + hB * 128 + wc * 32 + nt * 16 + (lane >> 4) * 4;
        if (gn >= N) continue;
        const float4 bv = (bias && !bias_done) ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int gm = m0 + hA * 128 + wr * 64 + mt * 16 + (lane & 15);
          if (gm >= M) continue;
          float v0 = acc[q][nt][mt][0] + bv.x, v1 = acc[q][nt][mt][1] + bv.y, v2 = acc[q][nt][mt][2] + bv.z,
                v3 = acc[q][nt][mt][3] + bv.w;
          if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
          if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
          if (EPI == EPI_BIAS_QGELU_BF16) { v0 = qgelu_fast(v0); v1 = qgelu_fast(v1); v2 = qgelu_fast(v2); v3 = qgelu_fast(v3); }
          if (EPI == EPI_BIAS_RESID_F32) {
            const float4 rv = ln_apply(*(const float4*)(resid + (size_t)gm * ldr + gn), ln, gm, gn);
            v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
          }
          if (F32_OUT) *(float4*)((float*)Cv + (size_t)gm * ldc + gn) = make_float4(v0, v1, v2, v3);
          else *(uint2*)((bf16_t*)Cv + (size_t)gm * ldc + gn) = make_uint2(pack2<DT>(v0, v1), pack2<DT>(v2, v3));
        }
      }
    }
  }
  stamp(stamps, 3);
}

// Variant HP ("persistent ring"): gemm_kernel_h with one workgroup per CU walking several output tiles.  Before a tile's
// epilogue starts, the first five half-tiles of the NEXT tile are already requested into ring slots 0-4 (free since the
// main loop's last barrier); the epilogue stages through the other half of the LDS ([80 KiB, 160 KiB), two 128-row passes
// for 16-bit output, four 64-row passes for fp32), so the 5-6k-cycle cold prologue of every tile but the first hides
// behind the previous tile's epilogue.  Same main loop, same arithmetic, same results as gemm_kernel_h.
// SPLIT (EPI_BIAS_RESID_F32 only): bit 0 = the residual rows come as the (hi, lo) pair, bit 1 = the output rows leave as one,
// bit 3 = the lo half (in and out) is the 8-bit e5m2 form of rr_common.h (GemmFold::lo_bits == 8) instead of fp16.
// DIAG (tools/bench_gemm.py --epilogue-timeline): 1 = per-wave s_memtime marks around the sections of the epilogue, summed over the
// workgroup's tiles; 2 = also the 13 marks per K-tile of the main loop (they cost the loop ~11 %).  0 in every product launch.
// FOLD: the folded-LayerNorm consumer form (ln.in_stats / ln.csum given) — a compile-time choice so that the accumulator
// arithmetic is one straight-line block (as a run-time branch its two arms joined in 128 accumulator phis and spilled).
template <int EPI, int DT, int SPLIT = 0, int DIAG = 0, bool FOLD = false>
__global__ __launch_bounds__(512) void gemm_kernel_hp(const bf16_t* __restrict__ A, int lda,
                                                     const bf16_t* __restrict__ W, int ldw,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ resid, int ldr,
                                                     void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                     int tiles_n, int nwg, unsigned long long* stamps, LnResid ln,
                                                     int stagger_unit) {
  // row panels per group of the tile order (below): 8 for K <= 1024, plain row-major for deeper K where one 256-row
  // activation panel is already 1.5 MB.  What the order sets is the L2 fill volume, not the speed: the 32 workgroups of an
  // XCD stream their K slices in step, a round of 32 tiles fetches (distinct row panels + distinct column slices) x 256 x K
  // x 2 bytes, and NOTHING survives in the 4 MB L2 from one round to the next (a round moves 4.7 MB at K = 768) — that
  // model reproduces FETCH_SIZE per kernel to 3-8 % (DESIGN.md "Round 3", profiles/r03_q_*).  8 row panels x 4 column
  // slices is the cheapest rectangle of 32; the step time is the same to 0.2 % for groups of 1 / 4 / 8
  // (profiles/r03_q_tile_order_ab.txt).
  const bool mrev = stagger_unit >= 1000;                   // this launch walks its tile lists from the end (rr_m_direction_next)
  if (mrev) stagger_unit -= 1000;
  // N <= 1024 (at most four column slices: the N = 768 residual GEMMs): plain row-major as well — the whole weight fits L2 anyway, and
  // the column tiles of a row panel then run side by side in ONE round instead of straddling two (attention-out reads 2.46 -> 2.13 GB
  // per launch beyond L2, profiles/r05_i_attn_out_tile_order.log; launch time within +-1 % for groups of 1 / 2 / 4 / 8,
  // profiles/r05_j_resid_tile_order.log: those re-reads are served by the memory-side cache and were never what the launch waits for).
  const int GROUP = stagger_unit >= 50 && stagger_unit <= 55 ? (2 << (stagger_unit - 50))      // A/B: 2..64
                    : (stagger_unit == 56 || Kd > 1024 || (tiles_n <= 4 && stagger_unit != 57)) ? 1 : 8;      // 57: the old rule (A/B)
  // De-synchronise the XCDs (rr_set_tuning "gemm_desync"; stagger_unit = 100 + u).  All workgroups of a persistent launch start
  // together and every tile costs the same, so the 256 CUs run their main loops (HBM nearly idle) and then their epilogues
  // (HBM saturated) in LOCKSTEP: tools/gemm_epilogue_timeline.py shows every workgroup inside its epilogue at the same
  // instant for the whole launch, and the residual epilogue moving its 134 MB per round at 6 TB/s — the HBM roofline —
  // while the main loops leave the memory idle.  The 32 workgroups of an XCD share operand slices through their L2, which
  // pulls stragglers back into step (a skew INSIDE the XCD did nothing: r03_a), so the skew is per XCD: the workgroups of
  // XCD x (blockIdx.x & 7 under round-robin placement; speed only, never correctness) start x * u * 256 cycles late, u
  // chosen by the host so that the eight XCDs cover a fraction of one tile period.
  if (stagger_unit >= 100) {
    const int n = (int)(blockIdx.x & 7) * (stagger_unit - 100);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);          // 4 * 64 cycles
  }
  stamp(stamps, 0);
  constexpr int BM = 256, BN = 256, HALF = 128 * 128;       // half-tile = 128 rows x 128 B
  // FOLD on a 16-bit / fp32-output GEMM: consumer side of the folded LayerNorm; on the fp32-stream residual GEMM (SPLIT 0):
  // producer side (x16 rows + statistics partials).  Compile-time, so that the residual GEMMs that emit nothing (fp8 mode,
  // plain residual launches) carry no reduction code and no registers for it.
  constexpr bool FOLD_IN = FOLD && EPI != EPI_BIAS_RESID_F32, FOLD_OUT = FOLD && EPI == EPI_BIAS_RESID_F32;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  // persistent XCD-aware walk: workgroup (xcd = bid & 7, j = bid >> 3) takes the tiles chunk0 + j + i * (grid / 8) of its
  // XCD's contiguous chunk of the tile list, i = 0, 1, ...
  const int bid = blockIdx.x, gstep = (int)(gridDim.x >> 3);
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, chunk_n = q8 + (xcd < r8 ? 1 : 0);
  int li = bid >> 3;
  if (li >= chunk_n) return;
  int m0, n0;
  // tile order inside the list: groups of GROUP row panels, column-major inside a group, so that the 32 tiles an XCD
  // works on at a time are ~4 weight column slices x 8 activation row panels (each W slice is re-used 8 times out of
  // L2 before it is evicted, instead of once per ~3 row panels with a row-major order)
  const int tiles_m = nwg / tiles_n;
  // Serpentine K walk: the tiles of every second block of four column slices take their K-tiles from the last to the first.
  // An XCD's round of 32 tiles is 8 row panels x 4 column slices walked in step; the next round has the same 8 panels and the
  // next 4 slices, and walking it backwards starts on the K slices of the panels that the round before touched LAST — the ones
  // the 4 MB L2 still holds (about 8 of the 12 at K = 768).  FETCH_SIZE per launch at 409 600 rows: FFN-up 2.83 -> 2.17 GB,
  // QKV (9 slices: rounds straddle the blocks) 2.47 -> 2.17 GB; FFN-up 1.981 -> 1.927 ms (profiles/r04_x_serpentine.log).  The
  // direction is a function of the column slice alone, so a row's arithmetic does not depend on where its tile lies (packed ==
  // padded, bit for bit); shapes of at most four slices (N <= 1024) are untouched.  rr_set_gemm_stagger(59) switches it off.
  const bool serp = stagger_unit != 59;
  int krev = 0;
  auto tile_origin = [&](int gidx, int& m0_, int& n0_) {
    const int per_group = GROUP * tiles_n, grp = gidx / per_group, r = gidx - grp * per_group;
    const int rows = min(GROUP, tiles_m - grp * GROUP);
    const int tn = r / rows, tm = grp * GROUP + (r - tn * rows);
    m0_ = tm * BM;
    n0_ = tn * BN;
    krev = serp ? k_walk_reversed(n0_) : 0;
  };
  tile_origin(chunk0 + (mrev ? chunk_n - 1 - li : li), m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- DMA sources: half-tile j in {0:A0, 1:B0, 2:B1, 3:A1}; this wave moves pieces 2*wave, 2*wave+1 (8 rows each).
  // 32-bit byte offsets from the scalar origin of the CURRENT tile's rows (a_tile / w_tile, re-based per output tile):
  // they span at most 256 row pitches, so A and W may be of any size.
  uint32_t so_a0[2], so_a1[2], so_b0[2], so_b1[2];
  const bf16_t *a_tile, *w_tile;
#define RR_SETUP_SRC(m0_, n0_)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
    const int r = (wave * 2 + i) * 8 + (lane >> 3);          /* row inside the half-tile */             \
    const int c = (lane & 7) ^ ((r >> 1) & 7);                                                          \
    so_a0[i] = (uint32_t)(((size_t)min(r, M - 1 - (m0_)) * lda + c * 8) * 2);                           \
    so_a1[i] = (uint32_t)(((size_t)min(128 + r, M - 1 - (m0_)) * lda + c * 8) * 2);                     \
    so_b0[i] = (uint32_t)(((size_t)min(r, N - 1 - (n0_)) * ldw + c * 8) * 2);                           \
    so_b1[i] = (uint32_t)(((size_t)min(128 + r, N - 1 - (n0_)) * ldw + c * 8) * 2);                     \
  }                                                                                                     \
  a_tile = A + (size_t)(m0_) * lda;                                                                     \
  w_tile = W + (size_t)(n0_) * ldw;
  RR_SETUP_SRC(m0, n0)
  const uint32_t lds_base = lds_addr(lds);
  const int nk = Kd / BK, H = 4 * nk;
  // half-tile (t, J): slot = parity (t&1) * 4 + J; J is a compile-time constant at every call site
#define RR_DMA(t_, J)                                                                                             \
  {                                                                                                               \
    const uint32_t dst_ = __builtin_amdgcn_readfirstlane(lds_base + (((t_) & 1) * 4 + (J)) * HALF + wave * 2048); \
    const int tk_ = krev ? nk - 1 - (t_) : (t_);             /* K-tile t of the walk is K-tile tk_ of the operands */         \
    const void* sb_ = ((J) == 0 || (J) == 3) ? (const void*)(a_tile + (size_t)tk_ * BK) : (const void*)(w_tile + (size_t)tk_ * BK); \
    const uint32_t* so_ = (J) == 0 ? so_a0 : (J) == 3 ? so_a1 : (J) == 1 ? so_b0 : so_b1;                         \
    glds16_so(sb_, so_[0], dst_);                                                                                 \
    glds16_so(sb_, so_[1], dst_ + 1024);                                                                          \
  }
  // wait until half-tile h_need (and everything older) has landed; h_last = newest half-tile issued so far
  auto wait_half = [&](int h_need, int h_last) {
    if (h_need >= H) return;
    const int after = h_last - h_need;                        // half-tiles issued after the needed one
    if (after >= 3) wait_vmcnt<6>();
    else if (after == 2) wait_vmcnt<4>();
    else if (after == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
  };

  // ---- fragment addresses inside a half-tile slot (swizzle term is the same for every 16-row block); the two
  // 32-deep k-steps of a fragment differ by chunk ^ 4 = byte offset ^ 64
  const int a_off = swz128(wr * 64 + (lane & 15), lane >> 4);
  const int b_off = swz128(wc * 32 + (lane & 15), lane >> 4);
  // DIAG, stagger_unit 62 / 63: every fragment read is issued twice / three times (the copies go to a scratch register quad):
  // what the main loop pays per EXTRA LDS read — i.e. what a wave tile with more operand reuse could win
  // (tools/gemm_epilogue_timeline.py --stagger 62).  Same operands, same MFMA work: only the LDS traffic changes.
  // (DIAG 3 = DIAG 1 + these reads: a build of its own, the loop around the copies costs the DIAG 1 timeline 18 % of its main loop)
  const int lds_dup = (DIAG == 3 && (stagger_unit == 62 || stagger_unit == 63)) ? stagger_unit - 61 : 0;     // 64: this build, no copies
  auto dup_read = [&](const char* p_) {
    if constexpr (DIAG == 3) {
      for (int d_ = 0; d_ < lds_dup; ++d_) {
        f32x4 scratch_;
        asm volatile("ds_read_b128 %0, %1" : "=v"(scratch_) : "v"(lds_addr(p_)) : "memory");
      }
    }
  };
  auto read_a = [&](const char* slot, int ks, bf16x8 (&f)[4]) {
    const char* b = slot + (ks ? (a_off ^ 64) : a_off);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { f[mt] = *(const bf16x8*)(b + mt * 2048); dup_read(b + mt * 2048); }
  };
  auto read_b = [&](const char* slot, int ks, bf16x8 (&f)[2]) {
    const char* b = slot + (ks ? (b_off ^ 64) : b_off);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { f[nt] = *(const bf16x8*)(b + nt * 2048); dup_read(b + nt * 2048); }
  };

  f32x4 acc[4][2][4];   // [quadrant 2*hA+hB][nt][mt]; lane: m = mt*16 + (lane&15), n = nt*16 + (lane>>4)*4 + reg

  // Register fragments: A of the current row half (both k-steps), B0 (kept p0..p3) and B1 (p1..p2).  Fragments for a
  // k-step are fetched one 8-MFMA block ahead of their use, so nothing is double-buffered: 64 fragment + 128
  // accumulator registers.
  bf16x8 AF0[4], AF1[4], B0K0[2], B0K1[2], B1K0[2], B1K1[2];

#define RR_BLK(Q, AF, BF)                                                                          \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)  \
      acc[Q][nt][mt] = mfma16<DT>(BF[nt], AF[mt], acc[Q][nt][mt]);
#define RR_SBAR() __builtin_amdgcn_sched_barrier(0)
  // wave priority falls with progress inside a barrier interval: of the two waves that share a SIMD the one that is
  // BEHIND wins the MFMA pipe, so they advance together instead of the older wave racing ahead to idle at the barrier
#define RR_PRIO(p) __builtin_amdgcn_s_setprio(p);

  // Per-tile epilogue parameters through LDS.  The accumulator arithmetic of the epilogue needs, per lane, the (mean, rstd) of
  // 8 rows (folded LayerNorm, consumer side) and bias / column sums of 32 columns.  As global loads they sat BEHIND the ten
  // LDS-DMA pieces of the next tile's prefetch in the wave's in-order vmcnt queue, so the first multiply-add of every tile
  // waited for 80 KiB of prefetch to land (tools/gemm_epilogue_timeline.py: 5.5-6.8k cycles per tile for 1-2k cycles of
  // arithmetic).  Now 16 dword LDS-DMA pieces per tile (2 per wave, exact per-dword clamping at ragged edges) bring them
  // to [152 KiB, 156 KiB) at the START of the tile's main loop — older than every ring refill of the tile, so the loop's
  // counted waits only get marginally stricter, never weaker — and the epilogue reads them with ds_read (lgkmcnt only).
  // Layout: +0 stats[256] float2, +2048 bias[256] f32, +3072 csum[256] f32; residual epilogue with the LayerNorm-recompute
  // (ln.stats): +4096 gamma[256], +5120 beta[256], +6144 (mean, rstd)[256] of the residual rows — 16 more pieces per tile —
  // so that its passes hold no per-thread gamma/beta registers across the tile and issue no statistics loads beside the
  // residual rows.  Same values, same arithmetic as before.
  constexpr int PARAM_OFF = 152 * 1024;
  auto stage_params = [&](int m0_, int n0_) {
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));     // opaque: the per-lane offsets below are recomputed here (a dozen VALU per tile), not hoisted out of the tile loop and spilled
    const int lane = lane_;
    if constexpr (FOLD_IN) {
      const int d = wave * 64 + lane;                                   // dword d of the 256 x float2 block
      glds4_so(ln.in_stats + m0_, (uint32_t)((min(d >> 1, M - 1 - m0_) * 2 + (d & 1)) * 4),
               __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + wave * 256));
    }
    if (wave < 4) {
      if (bias) glds4_so(bias + n0_, (uint32_t)(min(wave * 64 + lane, N - 1 - n0_) * 4),
                         __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 2048 + wave * 256));
    } else if (FOLD_IN) {
      glds4_so(ln.csum + n0_, (uint32_t)(min((wave - 4) * 64 + lane, N - 1 - n0_) * 4),
               __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 3072 + (wave - 4) * 256));
    }
    if (EPI == EPI_BIAS_RESID_F32 && ln.stats) {
      const int d = wave * 64 + lane;
      glds4_so(ln.stats + m0_, (uint32_t)((min(d >> 1, M - 1 - m0_) * 2 + (d & 1)) * 4),
               __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 6144 + wave * 256));
      glds4_so((wave < 4 ? ln.gamma : ln.beta) + n0_, (uint32_t)(min((wave & 3) * 64 + lane, N - 1 - n0_) * 4),
               __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 4096 + (wave >> 2) * 1024 + (wave & 3) * 256));
    }
  };
  if (!bias) {          // no bias vector: the block reads as zeros for the whole launch (made visible by the cold prologue's barrier)
    if (tid < 256) *(float*)(lds + PARAM_OFF + 2048 + tid * 4) = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // ---- first output tile: cold prologue, half-tiles 0..6 in flight (g = 4*tile + {A0:0, B0:1, B1:2, A1:3})
  stage_params(m0, n0);
  RR_DMA(0, 0) RR_DMA(0, 1) RR_DMA(0, 2) RR_DMA(0, 3)
  if (nk > 1) { RR_DMA(1, 0) RR_DMA(1, 1) RR_DMA(1, 2) }
  bool first_tile = true;
  // epilogue timeline (DIAG): ep[0] main loop, [1] next-tile setup + prefetch issue, [2] accumulator arithmetic, then per pass
  // summed: [3] staging writes, [4] prefetch confirm + touch, [5] barrier, [6] residual load issue, [7] residual load wait,
  // [8] stream-out body, [9] closing barrier; [10] tile tail, [11] tiles
  unsigned long long ep[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, em0 = 0, em1 = 0;
#define EP_MARK(var) { if constexpr (DIAG != 0) { RR_SBAR(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); RR_SBAR(); } }
#define EP_ADD(slot) { if constexpr (DIAG != 0) { EP_MARK(em1) ep[slot] += em1 - em0; em0 = em1; } }
  EP_MARK(em0)
  for (;;) {                                                // one iteration per output tile of this workgroup
  // First output tile: all of K-tile 0 landed (my pieces; the cold queue is [g0..g6], vmcnt(4) leaves g5, g6), then the
  // barrier.  Later tiles: every wave confirmed its pieces of the prefetched g0..g4 inside the previous epilogue — before
  // that epilogue's first store, so that the wait never covers a store acknowledgement — and has passed workgroup
  // barriers since: nothing to wait for here.
  if (first_tile) {
    if (nk > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }
  if (first_tile) stamp(stamps, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  read_a(lds + 0 * HALF, 0, AF0);
  read_b(lds + 1 * HALF, 0, B0K0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(AF0[0]), "+v"(AF0[1]), "+v"(AF0[2]), "+v"(AF0[3]), "+v"(B0K0[0]), "+v"(B0K0[1]));

  // Diagnostic build (DIAG): per-wave s_memtime marks inside every phase, summed over the loop and written to
  // stamps[block][8 + wave*8 + k]; the marks are read only after the phase's own lgkmcnt(0), so they add no wait.
  unsigned long long dg[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tmk[9];
#define RR_MARK(k) { if constexpr (DIAG == 2) { RR_SBAR(); asm volatile("s_memtime %0" : "=s"(tmk[k]) :: "memory"); RR_SBAR(); } }
#define RR_ACC(base, n)                                                                       \
  {                                                                                           \
    if constexpr (DIAG == 2) {                                                                \
      _Pragma("unroll") for (int k_ = 0; k_ < (n); ++k_) dg[(base) + k_] += tmk[k_ + 1] - tmk[k_]; \
    }                                                                                         \
  }
  // sync point with marks m0 (before), m0+1 (after the vmcnt wait), m0+2 (after lgkmcnt(0)), m0+3 (after the barrier).
  // In the steady state the wait is a literal (no scalar branch cascade in the loop body).
#define RR_SYNC(STEADY, NLIT, g_need, g_last, m0)        \
  {                                                      \
    RR_SBAR();                                           \
    RR_MARK(m0)                                          \
    if (STEADY) wait_vmcnt<NLIT>();                      \
    else wait_half(g_need, g_last);                      \
    RR_MARK((m0) + 1)                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    RR_MARK((m0) + 2)                                    \
    __builtin_amdgcn_s_barrier();                        \
    RR_SBAR();                                           \
    RR_MARK((m0) + 3)                                    \
  }

  // Two barriers per K-tile.  X (end of p1): A0/B0 of the next tile have landed and every wave is done with this
  // tile's A0, B0, B1 slots; Y (end of p3): B1/A1 of the next tile have landed and every wave is done with this tile's
  // A1 slot.  p0 and p2 end without a barrier: what they hand to p1 / p3 was covered by the previous Y / X.  (Four
  // barriers, one per phase, cost 17 % more main-loop cycles.)  Refills: p0 A1(t+1); p2 A0(t+2), B0(t+2); p3 B1(t+2) —
  // the g order the counted waits rely on.  Measured and NOT better: one barrier per tile (the refill window shrinks
  // to one tile and the drained DMA sets the pace), an LDS arrive/poll counter instead of s_barrier (s_barrier itself
  // is ~36 cycles, tools/barrier_probe.hip), one 1-KiB piece per 8-MFMA block instead of pairs (a piece costs its wave
  // ~60 cycles wherever it sits), all pieces moved by the younger wave of each SIMD pair under EXEC masking (-10 %).
  // RR_TILE(1) is the steady state (t <= nk-3): every refill exists and the vmcnt waits are literals, so the body is
  // straight-line code; RR_TILE(0) handles the last two tiles with the general guards.
#define RR_TILE(STEADY)                                                                                    \
  {                                                                                                        \
    const char* sl = lds + (t & 1) * 4 * HALF;               /* this tile's slots: +0 A0, +1 B0, +2 B1, +3 A1 */ \
    const char* sn = lds + ((t + 1) & 1) * 4 * HALF;         /* next tile's */                              \
    const bool d1 = ((STEADY) || t + 1 < nk) && !no_dma, d2 = ((STEADY) || t + 2 < nk) && !no_dma;          \
    /* ---- p0: quadrant (A0, B0) */                                                                        \
    RR_MARK(0)                                                                                             \
    read_a(sl + 0 * HALF, 1, AF1);                                                                         \
    read_b(sl + 1 * HALF, 1, B0K1);                                                                        \
    RR_SBAR();                                                                                             \
    RR_PRIO(3)                                                                                             \
    RR_BLK(0, AF0, B0K0)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_MARK(1)                                                                                             \
    read_b(sl + 2 * HALF, 0, B1K0);                                                                        \
    if (d1) RR_DMA(t + 1, 3)                                 /* A1 of the next tile (slot free since Y(t-1)) */ \
    RR_SBAR();                                                                                             \
    RR_MARK(2)                                                                                             \
    RR_PRIO(2)                                                                                             \
    RR_BLK(0, AF1, B0K1)                                                                                   \
    RR_SBAR();                                                                                             \
    /* ---- p1: quadrant (A0, B1) */                                                                        \
    RR_MARK(3)                                                                                             \
    read_b(sl + 2 * HALF, 1, B1K1);                                                                        \
    RR_SBAR();                                                                                             \
    RR_PRIO(1)                                                                                             \
    RR_BLK(1, AF0, B1K0)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_MARK(4)                                                                                             \
    read_a(sl + 3 * HALF, 0, AF0);                                                                         \
    RR_SBAR();                                                                                             \
    RR_PRIO(0)                                                                                             \
    RR_BLK(1, AF1, B1K1)                                                                                   \
    RR_SYNC(STEADY, 4, 4 * (t + 1) + 1, min(H - 1, 4 * (t + 1) + 3), 5)   /* X: A0(t+1), B0(t+1) landed */    \
    RR_ACC(0, 8)                                                                                           \
    /* ---- p2: quadrant (A1, B1) */                                                                        \
    RR_MARK(0)                                                                                             \
    read_a(sl + 3 * HALF, 1, AF1);                                                                         \
    if (d2) RR_DMA(t + 2, 0)                                 /* slots A0, B0 (free since X) */               \
    RR_SBAR();                                                                                             \
    RR_PRIO(3)                                                                                             \
    RR_BLK(3, AF0, B1K0)                                                                                   \
    RR_SBAR();                                                                                             \
    if (d2) RR_DMA(t + 2, 1)                                                                               \
    RR_SBAR();                                                                                             \
    RR_PRIO(2)                                                                                             \
    RR_BLK(3, AF1, B1K1)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_MARK(1)                                                                                             \
    /* ---- p3: quadrant (A1, B0) */                                                                        \
    RR_PRIO(1)                                                                                             \
    RR_BLK(2, AF0, B0K0)                                                                                   \
    RR_SBAR();                                                                                             \
    if ((STEADY) || t + 1 < nk) {                                                                          \
      read_a(sn + 0 * HALF, 0, AF0);                                                                       \
      read_b(sn + 1 * HALF, 0, B0K0);                                                                      \
    }                                                                                                      \
    if (d2) RR_DMA(t + 2, 2)                                 /* slot B1 (free since X) */                    \
    RR_SBAR();                                                                                             \
    RR_PRIO(0)                                                                                             \
    RR_BLK(2, AF1, B0K1)                                                                                   \
    RR_SYNC(STEADY, 6, 4 * (t + 1) + 3, min(H - 1, 4 * (t + 2) + 2), 2)   /* Y: B1(t+1), A1(t+1) landed */    \
    /* The next tile's first fragments are complete here (lgkmcnt(0) above).  Tell the compiler: otherwise it    \
       treats them as pending across the back edge and puts lgkmcnt(0) behind the six ds_reads that open p0. */  \
    asm volatile("" : "+v"(AF0[0]), "+v"(AF0[1]), "+v"(AF0[2]), "+v"(AF0[3]), "+v"(B0K0[0]), "+v"(B0K0[1])); \
    RR_ACC(8, 5)                                                                                           \
  }
  const bool no_dma = DIAG == 2 && stagger_unit == 61;            // diagnostic: main loop without refills (wrong results)
  int t = 0;
  for (; t < nk - 2; ++t) RR_TILE(1)
  for (; t < nk; ++t) RR_TILE(0)
#undef RR_TILE
  if constexpr (DIAG == 2) {
    if (stamps && lane == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned long long* o = stamps + (size_t)gridDim.x * 8 + ((size_t)blockIdx.x * 8 + wave) * 16;
      for (int k = 0; k < 13; ++k) o[k] = dg[k];
    }
  }
#undef RR_SYNC
#undef RR_MARK
#undef RR_ACC
  wait_vmcnt<0>();   // nothing is in flight any more (every issued half-tile was waited for); explicit before LDS reuse
  // One K-tile only (K = 64): the loop above has no counted wait at all (wait_half returns for half-tiles beyond the last), so its
  // barriers did not order the OTHER waves' parameter-block pieces — requested behind the previous tile's epilogue — before this
  // tile's epilogue reads them: a later tile of a workgroup could add the previous column tile's bias (found in round 5 by
  // test_ring_kernels_ragged_multi_tile[8453-2304-64-14], intermittent).  With two or more K-tiles the first counted wait + barrier
  // of the loop covers the block, which is older than every ring refill of its tile.  Uniform branch, never taken on the path (K >= 128).
  if (nk == 1) __builtin_amdgcn_s_barrier();
  if (first_tile) stamp(stamps, 2);
  EP_ADD(0)
  // DIAG: wall-clock (100 MHz s_memrealtime) of the start and end of the first 32 epilogues of every workgroup, to see
  // whether the CUs run their epilogues in lockstep (stamps + grid * (8 + 256) + block * 64 + 2 * tile)
  if constexpr (DIAG != 0) {
    if (stamps && tid == 0 && ep[11] < 32)
      stamps[(size_t)gridDim.x * 264 + (size_t)blockIdx.x * 64 + 2 * ep[11]] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- next output tile of this workgroup: request its first five half-tiles into ring slots 0-4 now (every slot has
  // been free since the last barrier of the main loop); the epilogue below works in the upper LDS half only
  const int cm0 = m0, cn0 = n0;                             // the tile being written out
  li += gstep;
  const bool has_next = li < chunk_n;
  if (has_next) {
    tile_origin(chunk0 + (mrev ? chunk_n - 1 - li : li), m0, n0);
    RR_SETUP_SRC(m0, n0)
    RR_DMA(0, 0) RR_DMA(0, 1) RR_DMA(0, 2) RR_DMA(0, 3)
    if (nk > 1) RR_DMA(1, 0)
  }
  EP_ADD(1)

  // ---- epilogue of tile (cm0, cn0), staged through the UPPER LDS half [5 * HALF, 160 KiB): 16-bit output in two 128-row
  // passes (pass = hA), fp32 in four 64-row passes (pass = 2*hA + wr: one wave row group at a time)
  constexpr bool F32_OUT = (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RESID_F32);
  {
    constexpr int ES = F32_OUT ? 4 : 2;
    constexpr int PITCH = BN * ES + 16;
    constexpr int CPR = BN * ES / 16;
    constexpr int NPASS = F32_OUT ? 4 : 2;
    constexpr int ROWS = F32_OUT ? 64 : 128;
    static_assert(ROWS * PITCH <= 5 * HALF, "staging image must fit above the five prefetch slots");
    char* const stg = lds + 5 * HALF;
    // The epilogue's per-lane addresses are recomputed from an OPAQUE thread index once per tile (a few dozen VALU): left
    // visible, hipcc hoists them out of the tile loop as loop invariants, keeps them live across the main loop at 256
    // VGPRs and spills them — and a scratch reload is a vector-memory operation that queues behind the prefetch DMA.
    int tid_o_ = tid;
    asm volatile("" : "+v"(tid_o_));
    const int tid = tid_o_, lane = tid_o_ & 63;
    // Accumulator arithmetic.  Which form runs (folded LayerNorm or plain bias) is decided ONCE per tile, and bias / column
    // sums are read from the parameter block without a column test (the block always holds 256 entries: beyond N the DMA
    // repeated the last column, whose results are never stored; without a bias the block was zeroed at kernel start).  The
    // first version tested `ln.in_stats`, `bias` and `gn < N` around every group of four multiply-adds: 100+ scalar
    // branches and exec-mask regions per tile (3.7-3.9k cycles for ~1k cycles of arithmetic in the epilogue timeline).
    auto activate = [&](f32x4 ab) -> f32x4 {
      float v0 = ab[0], v1 = ab[1], v2 = ab[2], v3 = ab[3];
      if constexpr (EPI == EPI_BIAS_GELU_BF16 && RR_PK_GELU != 0) {
        const f32x2 g0 = gelu_erf_fast2(f32x2{v0, v1}), g1 = gelu_erf_fast2(f32x2{v2, v3});
        return f32x4{g0.x, g0.y, g1.x, g1.y};
      }
      if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
      if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
      if (EPI == EPI_BIAS_QGELU_BF16) { v0 = qgelu_fast(v0); v1 = qgelu_fast(v1); v2 = qgelu_fast(v2); v3 = qgelu_fast(v3); }
      return f32x4{v0, v1, v2, v3};
    };
    const char* const pcol = lds + PARAM_OFF + 2048 + (wc * 32 + (lane >> 4) * 4) * 4;     // bias of this lane's first column; csum at +1024
    // The parameter reads run one (quadrant, column block) step ahead of the arithmetic that consumes them, pinned with
    // scheduling fences: left free, hipcc hoists all 16 ds_read_b128 (64 registers) to the top of the section, which at
    // 128 live accumulators spills the main loop's carried registers around it.
    if constexpr (FOLD_IN) {
      // folded LayerNorm, consumer side: A held raw pre-LayerNorm rows; (mean, rstd) of the 8 rows this lane owns
      float2 fst[2][4];
#pragma unroll
      for (int hA = 0; hA < 2; ++hA)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          fst[hA][mt] = *(const float2*)(lds + PARAM_OFF + (hA * 128 + wr * 64 + mt * 16 + (lane & 15)) * 8);
      float4 bv_n = *(const float4*)(pcol), cs_n = *(const float4*)(pcol + 1024);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it >> 1, nt = it & 1;
        const float4 bv = bv_n, cs = cs_n;
        if (it + 1 < 8) {
          const int o = ((((it + 1) >> 1) & 1) * 128 + ((it + 1) & 1) * 16) * 4;
          bv_n = *(const float4*)(pcol + o);
          cs_n = *(const float4*)(pcol + 1024 + o);
        }
        RR_SBAR();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[q][nt][mt] = activate(fold_apply(acc[q][nt][mt], fst[q >> 1][mt], cs, bv));
        RR_SBAR();
      }
    } else {
      float4 bv_n = *(const float4*)(pcol);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it >> 1, nt = it & 1;
        const float4 bv = bv_n;
        if (it + 1 < 8) bv_n = *(const float4*)(pcol + ((((it + 1) >> 1) & 1) * 128 + ((it + 1) & 1) * 16) * 4);
        RR_SBAR();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[q][nt][mt] = activate(f32x4{acc[q][nt][mt][0] + bv.x, acc[q][nt][mt][1] + bv.y, acc[q][nt][mt][2] + bv.z, acc[q][nt][mt][3] + bv.w});
        RR_SBAR();
      }
    }
    EP_ADD(2)
    const int my_col = (EPI == EPI_BIAS_RESID_F32 && SPLIT != 0) ? cn0 + (tid & 31) * 8 : cn0 + (tid % CPR) * (16 / ES);
    // gamma / beta of this thread's columns: from the parameter block, re-read in every pass (two or four ds_read_b128)
    auto load_gb = [&](float4& lg, float4& lb, float4& lg1, float4& lb1) {
      lg = make_float4(1.f, 1.f, 1.f, 1.f); lb = make_float4(0.f, 0.f, 0.f, 0.f); lg1 = lg; lb1 = lb;   // lg1 / lb1: split path, columns +4..+7
      if (EPI == EPI_BIAS_RESID_F32 && ln.stats && my_col < N) {
        const char* pp = lds + PARAM_OFF + 4096 + (my_col - cn0) * 4;
        lg = *(const float4*)pp;
        lb = *(const float4*)(pp + 1024);
        if constexpr (SPLIT != 0) { lg1 = *(const float4*)(pp + 16); lb1 = *(const float4*)(pp + 1024 + 16); }
      }
    };
    // The residual rows of a pass are first touched by a register-free LDS-DMA, one 128-byte line per thread (64 rows x
    // 1 KiB = 512 lines), a pass ahead of the loads that consume them: the loads behind the barrier then find their lines
    // in the XCD's L2 instead of paying the HBM latency in front of the first add, four times per tile.  The touched
    // dwords land in the unused tail of the staging region and are never read.
    auto touch_resid = [&](int pass_) {
      if (EPI != EPI_BIAS_RESID_F32 || !(ln.flags & 1)) return;
      if (cm0 + 256 > M || cn0 + 256 > N) return;                   // ragged last tiles: not worth a clamp per lane
      const size_t row0 = (size_t)(cm0 + (pass_ >> 1) * 128 + (pass_ & 1) * 64);
      if constexpr ((SPLIT & 1) != 0) {   // split residual: 64 rows x 512 B of hi (waves 0-3) and of lo (waves 4-7), one 128-byte line per thread
        if constexpr ((SPLIT & 8) != 0) {   // 8-bit lo: 64 rows x 256 B = 128 lines (waves 4-5)
          if (wave < 4) {
            const bf16_t* sb = ln.r_hi + row0 * ln.ld16 + cn0;                                          // scalar
            const uint32_t vo = (uint32_t)((((tid & 255) >> 2) * ln.ld16 + (tid & 3) * 64) * 2);
            glds4_so(sb, vo, __builtin_amdgcn_readfirstlane(lds_base + 5 * HALF + ROWS * PITCH + wave * 256));
          } else if (wave < 6) {             // the pass's 32 row PAIRS (lo8_pair_offset) x 512 B
            const uint8_t* sb = (const uint8_t*)ln.r_lo + (row0 >> 5) * 16 * (size_t)(2 * ln.ld16) + cn0 * 2;   // scalar
            const uint32_t vo = (uint32_t)(((tid & 127) >> 2) * (2 * ln.ld16) + (tid & 3) * 128);
            glds4_so(sb, vo, __builtin_amdgcn_readfirstlane(lds_base + 5 * HALF + ROWS * PITCH + wave * 256));
          }
        } else {
        const bf16_t* sb = (wave < 4 ? ln.r_hi : ln.r_lo) + row0 * ln.ld16 + cn0;                     // scalar
        const uint32_t vo = (uint32_t)((((tid & 255) >> 2) * ln.ld16 + (tid & 3) * 64) * 2);
        glds4_so(sb, vo, __builtin_amdgcn_readfirstlane(lds_base + 5 * HALF + ROWS * PITCH + wave * 256));
        }
      } else {
        const float* sb = resid + row0 * ldr + cn0;                                                     // scalar
        const uint32_t vo = (uint32_t)(((tid >> 3) * ldr + (tid & 7) * 32) * 4);                        // line `tid` of the 64 x 256 block
        glds4_so(sb, vo, __builtin_amdgcn_readfirstlane(lds_base + 5 * HALF + ROWS * PITCH + wave * 256));
      }
    };
    static_assert(5 * HALF + ROWS * PITCH + (F32_OUT ? 8 * 256 : 0) <= PARAM_OFF, "staging image + touch area must end below the parameter block");
    touch_resid(0);
    // ---- split residual stream: the residual rows of pass p + 1 are requested at the END of pass p's body, into the registers
    // pass p has just consumed, so that their latency runs under the closing barrier, the next staging writes and the barrier
    // behind them instead of in front of the first add of every pass (epilogue timeline: load issue + wait were 40 % of the
    // residual epilogue).  Pass 0's are requested here.  In-place use stays safe: a pass reads and writes only its own rows.
    constexpr int UN8 = 4;
    const int c8 = tid & 31, gcol8 = cn0 + c8 * 8;
    const bool col_ok8 = gcol8 < N;                                   // N % 8 == 0: a chunk is inside or outside as a whole
    constexpr bool SPLIT_EPI = (EPI == EPI_BIAS_RESID_F32 && SPLIT != 0);
    constexpr bool LO8 = SPLIT_EPI && (SPLIT & 8) != 0;
    constexpr float LO8_MX = 1.0f / (float)(1 << RR_LO8_SHIFT);      // the MX scale of the e5m2 lo bytes (rr_common.h)
    uint4 rh[SPLIT_EPI ? UN8 : 1], rl[(SPLIT_EPI && !LO8) ? UN8 : 1];
    uint4 rl8[LO8 ? UN8 / 2 : 1];                                      // one 16-byte chunk = 8 columns of rows r and r + 16 (lo8_pair_offset)
    float4 ra[(SPLIT_EPI && !(SPLIT & 1)) ? UN8 : 1], rb[(SPLIT_EPI && !(SPLIT & 1)) ? UN8 : 1];
    auto issue_resid = [&](int pass_) {
      if constexpr (SPLIT_EPI) {
        const int rbase_ = cm0 + (pass_ >> 1) * 128 + (pass_ & 1) * 64;
#pragma unroll
        for (int u = 0; u < UN8; ++u) {
          const int gm = rbase_ + (tid >> 5) + u * 16;
          const bool ok = gm < M && col_ok8;
          if constexpr ((SPLIT & 1) != 0 && (RR_EPI_DIAG & 2) != 0) {
            rh[u] = make_uint4(0x3c003c00u + gm, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
            rl[u] = make_uint4(0x10001000u + tid, 0x10001000u, 0x10001000u, 0x10001000u);
          } else if constexpr ((SPLIT & 1) != 0 && LO8) {
            rh[u] = ok ? load_stream_u4(ln.r_hi + (size_t)gm * ln.ld16 + gcol8) : make_uint4(0u, 0u, 0u, 0u);
            if (!(u & 1)) rl8[u >> 1] = ok ? load_stream_u4((const uint8_t*)ln.r_lo + lo8_pair_offset(gm, gcol8, ln.ld16)) : make_uint4(0u, 0u, 0u, 0u);
          } else if constexpr ((SPLIT & 1) != 0) {
            rh[u] = ok ? load_stream_u4(ln.r_hi + (size_t)gm * ln.ld16 + gcol8) : make_uint4(0u, 0u, 0u, 0u);
            if constexpr ((RR_EPI_DIAG & 8) != 0) rl[u] = (ok && !(u & 1)) ? load_stream_u4(ln.r_lo + (size_t)gm * ln.ld16 + gcol8) : make_uint4(0u, 0u, 0u, 0u);
            else rl[u] = ok ? load_stream_u4(ln.r_lo + (size_t)gm * ln.ld16 + gcol8) : make_uint4(0u, 0u, 0u, 0u);
          } else {
            ra[u] = ok ? load_stream_f4(resid + (size_t)gm * ldr + gcol8) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[u] = ok ? load_stream_f4(resid + (size_t)gm * ldr + gcol8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    };
    issue_resid(0);
    EP_ADD(6)
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hA = q >> 1, hB = q & 1;
        if (F32_OUT ? (hA != (pass >> 1) || wr != (pass & 1)) : (hA != pass)) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int cn = hB * 128 + wc * 32 + nt * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int r = (F32_OUT ? 0 : wr * 64) + mt * 16 + (lane & 15);
            // 16-bit image: rows r and r+8 of a 16-lane group would meet on the same banks (528-B pitch = 4 banks per row,
            // 8 bytes per lane: 2-way, 8.7 % of this kernel's LDS cycles by SQ_LDS_BANK_CONFLICT); rows with bit 3 set
            // keep the two 8-byte halves of every 16-byte chunk swapped, the reader swaps them back
            char* dst = stg + r * PITCH + (F32_OUT ? cn * ES : ((cn * ES) ^ (lane & 8)));
            if constexpr (EPI == EPI_BIAS_RESID_F32 && SPLIT != 0)       // even 16-byte chunks | odd 16-byte chunks (see the stream-out)
              dst = stg + r * PITCH + (((cn >> 3) << 4) | (((cn >> 2) & 1) << 9));
            if constexpr (F32_OUT) *(f32x4*)dst = acc[q][nt][mt];
            else *(uint2*)dst = make_uint2(pack2<DT>(acc[q][nt][mt][0], acc[q][nt][mt][1]),
                                           pack2<DT>(acc[q][nt][mt][2], acc[q][nt][mt][3]));
          }
        }
      }
      EP_ADD(3)
      // the next tile's prefetched half-tiles: confirm MY pieces now, while the only younger vector-memory operations
      // are this epilogue's own loads (none issued yet in this pass) — not after the stores
      if (pass == 0 && has_next) wait_vmcnt<0>();
      if (pass + 1 < NPASS) touch_resid(pass + 1);
      EP_ADD(4)
      lds_barrier();
      EP_ADD(5)
      const int row_base = cm0 + (F32_OUT ? (pass >> 1) * 128 + (pass & 1) * 64 : pass * 128);
      if constexpr (EPI == EPI_BIAS_RESID_F32 && SPLIT != 0) {
        // ---- split residual stream (GemmFold in rr_common.h).  A thread owns 4 chunks of EIGHT columns per pass, so that the
        // 16-bit rows move as 16 bytes per lane (8-byte accesses reach 0.54-0.70x of the 16-byte rate: the first version of
        // this path, with the 4-column chunks of the fp32 form, moved 20 % fewer bytes and was 0.5 % slower).  A wave covers
        // two rows per step (lanes 0-31 / 32-63), a 128-column group is one DPP row of 16 lanes.  The staging image keeps
        // the even 16-byte chunks of a row in its first 512 bytes and the odd ones in the second, so that the two reads
        // of a lane are each 16 consecutive bytes per lane across a 16-lane group (conflict-free).
        static_assert(ROWS * 32 == 512 * UN8, "one batch of four 8-column chunks per thread and pass");
        const int gcol = gcol8;
        const bool col_ok = col_ok8;
        float2 rst[UN8];
#pragma unroll
        for (int u = 0; u < UN8; ++u) {
          const int gm = row_base + (tid >> 5) + u * 16;
          rst[u] = ln.stats ? *(const float2*)(lds + PARAM_OFF + 6144 + (gm - cm0) * 8) : make_float2(0.f, 1.f);
        }
        float4 lg, lb, lg1, lb1;
        load_gb(lg, lb, lg1, lb1);
        if constexpr (DIAG != 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        uint32_t l8_even[2] = {0u, 0u};      // 8-bit lo: the bytes of the even row of a pair wait here for the odd row's (one 16-byte store per pair)
        EP_ADD(7)
#pragma unroll
        for (int u = 0; u < UN8; ++u) {
          const int r = (tid >> 5) + u * 16, gm = row_base + r;
          const bool ok = gm < M && col_ok;
          float x[8];
          if constexpr ((SPLIT & 1) != 0 && LO8) {   // x = hi + e5m2 lo: one scaled unpack per pair, one v_fma_mix_f32 / v_add_f32 per element
            const uint32_t tok = mix_fence(rh[u], rl8[u >> 1]);
            const uint32_t hw[4] = {rh[u].x, rh[u].y, rh[u].z, rh[u].w};
            const uint32_t w0 = (u & 1) ? rl8[u >> 1].z : rl8[u >> 1].x, w1 = (u & 1) ? rl8[u >> 1].w : rl8[u >> 1].y;
            const float2 l2[4] = {lo8_decode<0>(w0, LO8_MX, tok), lo8_decode<1>(w0, LO8_MX, tok),
                                  lo8_decode<0>(w1, LO8_MX, tok), lo8_decode<1>(w1, LO8_MX, tok)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if constexpr (DT == 1) {
                x[2 * j] = mix_add_f16_f32<0>(l2[j].x, hw[j], tok); x[2 * j + 1] = mix_add_f16_f32<1>(l2[j].y, hw[j], tok);
              } else {
                const float2 h = unpack2<0>(hw[j]);
                x[2 * j] = h.x + l2[j].x; x[2 * j + 1] = h.y + l2[j].y;
              }
            }
          } else if constexpr ((SPLIT & 1) != 0) {      // x = hi (operand type) + lo (fp16): one v_fma_mix_f32 per element (rr_common.h)
            const uint32_t tok = mix_fence(rh[u], rl[u]);
            const uint32_t hw[4] = {rh[u].x, rh[u].y, rh[u].z, rh[u].w}, lw[4] = {rl[u].x, rl[u].y, rl[u].z, rl[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if constexpr (DT == 1) {
                x[2 * j] = mix_add_f16<0>(hw[j], lw[j], tok); x[2 * j + 1] = mix_add_f16<1>(hw[j], lw[j], tok);
              } else {
                const float2 h = unpack2<0>(hw[j]);
                x[2 * j] = mix_add_f16_f32<0>(h.x, lw[j], tok); x[2 * j + 1] = mix_add_f16_f32<1>(h.y, lw[j], tok);
              }
            }
          } else {
            x[0] = ra[u].x; x[1] = ra[u].y; x[2] = ra[u].z; x[3] = ra[u].w;
            x[4] = rb[u].x; x[5] = rb[u].y; x[6] = rb[u].z; x[7] = rb[u].w;
          }
          if (ln.stats) {
            const float ga[8] = {lg.x, lg.y, lg.z, lg.w, lg1.x, lg1.y, lg1.z, lg1.w};
            const float ba[8] = {lb.x, lb.y, lb.z, lb.w, lb1.x, lb1.y, lb1.z, lb1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (x[j] - rst[u].x) * rst[u].y * ga[j] + ba[j];
          }
          const float4 v0 = *(const float4*)(stg + r * PITCH + c8 * 16);           // columns gcol .. gcol+3
          const float4 v1 = *(const float4*)(stg + r * PITCH + 512 + c8 * 16);     // columns gcol+4 .. gcol+7
          float f[8] = {v0.x + x[0], v0.y + x[1], v0.z + x[2], v0.w + x[3], v1.x + x[4], v1.y + x[5], v1.z + x[6], v1.w + x[7]};
          if (!ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
          }
          uint32_t hi[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) hi[j] = pack2<DT>(f[2 * j], f[2 * j + 1]);
          if constexpr (LO8 && (SPLIT & 2) != 0) {      // row u - 1 was inside the matrix, its partner row u is not: its chunk leaves with a zero second half
            if ((u & 1) && !ok && gm - 16 < M && col_ok)
              store_stream((uint8_t*)ln.lo_out + lo8_pair_offset(gm - 16, gcol, ln.ld16), make_uint4(l8_even[0], l8_even[1], 0u, 0u));
          }
          bool st_ok = ok;
          if constexpr ((RR_EPI_DIAG & 1) != 0) st_ok = ok && f[0] == 1.2345e38f;
          if (st_ok) {
            if constexpr (SPLIT != 4) store_stream(ln.x16 + (size_t)gm * ln.ldx + gcol, make_uint4(hi[0], hi[1], hi[2], hi[3]));
            if constexpr ((SPLIT & 2) != 0 && LO8) {    // lo = e5m2((x - hi) * 2^RR_LO8_SHIFT): x - hi is exact in fp32, one rounding
              uint32_t l8[2] = {0u, 0u};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float d0, d1;
                if constexpr (DT == 1) {
                  d0 = mix_sub_f16<0>(f[2 * j], hi[j]); d1 = mix_sub_f16<1>(f[2 * j + 1], hi[j]);
                } else {
                  const float2 hb = unpack2<0>(hi[j]);
                  d0 = f[2 * j] - hb.x; d1 = f[2 * j + 1] - hb.y;
                }
                if (j & 1) l8[j >> 1] = lo8_encode<1>(l8[j >> 1], d0, d1, LO8_MX);
                else l8[j >> 1] = lo8_encode<0>(l8[j >> 1], d0, d1, LO8_MX);
              }
              if (u & 1) store_stream((uint8_t*)ln.lo_out + lo8_pair_offset(gm - 16, gcol, ln.ld16), make_uint4(l8_even[0], l8_even[1], l8[0], l8[1]));
              else { l8_even[0] = l8[0]; l8_even[1] = l8[1]; }
            } else if constexpr ((SPLIT & 2) != 0) {    // lo = fp16(x - hi)
              uint32_t lo[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if constexpr (DT == 1) {
                  lo[j] = pack2<1>(mix_sub_f16<0>(f[2 * j], hi[j]), mix_sub_f16<1>(f[2 * j + 1], hi[j]));
                } else {
                  const float2 hb = unpack2<0>(hi[j]);
                  lo[j] = pack2<1>(f[2 * j] - hb.x, f[2 * j + 1] - hb.y);
                }
              }
              if (!((RR_EPI_DIAG & 8) != 0 && (u & 1)))
                store_stream(ln.lo_out + (size_t)gm * ln.ld16 + gcol, make_uint4(lo[0], lo[1], lo[2], lo[3]));
            } else {
              float* cp = (float*)Cv + (size_t)gm * ldc + gcol;
              store_stream(cp, make_float4(f[0], f[1], f[2], f[3]));
              store_stream(cp + 4, make_float4(f[4], f[5], f[6], f[7]));
            }
          }
          // LayerNorm statistics of the 128-column group = the 16 lanes of this DPP row (every lane takes part)
          if constexpr (SPLIT != 4 && (RR_EPI_DIAG & 4) == 0) {           // (SPLIT 4: the plain fp32 stream on this epilogue — no 16-bit copy, no statistics)
            const int grp = gcol >> 7;
            const float rcnt = __builtin_amdgcn_rcpf((float)max(1, min(128, N - (grp << 7))));
            const float mg = row16_sum(((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]))) * rcnt;
            float q = 0.f;
            if (ok) {
#pragma unroll
              for (int j = 0; j < 8; ++j) q += (f[j] - mg) * (f[j] - mg);
            }
            const float m2 = row16_sum(q);
            if (st_ok && (tid & 15) == 0 && ((RR_EPI_DIAG & 16) == 0 || mg == 1.2345e38f)) ln.part[(size_t)gm * ln.nparts + grp] = make_float2(mg, m2);
          }
        }
        EP_ADD(8)
        if (pass + 1 < NPASS) issue_resid(pass + 1);
        EP_ADD(6)
      } else {
      // the whole pass in one batch of 8 sixteen-byte chunks per thread: every residual load is issued before the first
      // add/store (a wave keeps one row per step: its LayerNorm statistics are a scalar load)
      constexpr int UNR = 8;
      static_assert(512 % CPR == 0 && ROWS * CPR == 512 * UNR, "one batch per pass; a wave must not straddle rows");
      if constexpr (EPI != EPI_BIAS_RESID_F32) {
        // all eight staged chunks are read first (unconditionally: the image always holds the whole pass), then stored:
        // read-wait-store per chunk exposed one LDS latency eight times per pass
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint4 sv[UNR / 2];
#pragma unroll
          for (int u = 0; u < UNR / 2; ++u) {
            const int i = tid + (h * (UNR / 2) + u) * 512, r = i / CPR, c = i - r * CPR;
            sv[u] = *(const uint4*)(stg + r * PITCH + c * 16);
          }
          const bool sw = !F32_OUT && (tid & 256);                // r = tid/32 + 16u: bit 3 of r = bit 8 of tid
#pragma unroll
          for (int u = 0; u < UNR / 2; ++u) {
            const int i = tid + (h * (UNR / 2) + u) * 512, r = i / CPR, c = i - r * CPR;
            const int gm = row_base + r, gcol = cn0 + c * (16 / ES);
            const uint4 v = make_uint4(sw ? sv[u].z : sv[u].x, sw ? sv[u].w : sv[u].y, sw ? sv[u].x : sv[u].z, sw ? sv[u].y : sv[u].w);
            if (gm < M && gcol < N) store_stream((char*)Cv + ((size_t)gm * ldc + gcol) * ES, v);
          }
          RR_SBAR();
        }
      } else {
      // fp32 residual stream (first / last layer of a stack, fp8 mode): two half-batches of four chunks, the four residual
      // loads of a half issued before its first add (eight at once spilled 45 registers around the pass at 96 live accumulators)
      float4 lg, lb, lg1, lb1;
      load_gb(lg, lb, lg1, lb1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        constexpr int UH = UNR / 2;
        float4 rv[UH];
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int i = tid + (h * UH + u) * 512, r = i / CPR, c = i - r * CPR;
          const int gm = row_base + r, gcol = cn0 + c * (16 / ES);
          const bool ok = gm < M && gcol < N;
          float4 x = ok ? load_stream_f4(resid + (size_t)gm * ldr + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (ln.stats) {
            const float2 st2 = *(const float2*)(lds + PARAM_OFF + 6144 + (gm - cm0) * 8);   // (rows beyond M were clamped to M - 1 by the DMA)
            x = make_float4((x.x - st2.x) * st2.y * lg.x + lb.x, (x.y - st2.x) * st2.y * lg.y + lb.y,
                            (x.z - st2.x) * st2.y * lg.z + lb.z, (x.w - st2.x) * st2.y * lg.w + lb.w);
          }
          rv[u] = x;
        }
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int i = tid + (h * UH + u) * 512, r = i / CPR, c = i - r * CPR;
          const int gm = row_base + r, gcol = cn0 + c * (16 / ES);
          const bool ok = gm < M && gcol < N;
          float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) {
            f = *(const float4*)(stg + r * PITCH + c * 16);
            f.x += rv[u].x; f.y += rv[u].y; f.z += rv[u].z; f.w += rv[u].w;
            store_stream((char*)Cv + ((size_t)gm * ldc + gcol) * ES, f);
          }
          // folded LayerNorm, producer side: the 16-bit copy of the row and its statistics per 128-column group (a wave holds
          // one row of this tile per step, lane = 16-byte chunk: lanes 0-31 / 32-63 are the tile's two column groups)
          if constexpr (FOLD_OUT) { if (ln.x16) fold_emit<DT>(ln, f, ok, gm, gcol, N); }
        }
        RR_SBAR();
      }
      }
      }
      EP_ADD(8)
      lds_barrier();                                       // staging image consumed (next pass / next tile may overwrite it)
      EP_ADD(9)
    }
  }
  if (first_tile) stamp(stamps, 3);
  first_tile = false;
  if constexpr (DIAG != 0) {
    if (stamps && tid == 0 && ep[11] < 32)
      stamps[(size_t)gridDim.x * 264 + (size_t)blockIdx.x * 64 + 2 * ep[11] + 1] = __builtin_amdgcn_s_memrealtime();
    ep[11] += 1;
  }
  if (!has_next) break;
  stage_params(m0, n0);                                     // the NEXT tile's (this tile's epilogue has read its own)
  if (nk > 1) { RR_DMA(1, 1) RR_DMA(1, 2) }                 // slots 5, 6 were under the staging image until now
  EP_ADD(10)
  }   // output tiles
  if constexpr (DIAG != 0) {
    if (stamps && lane == 0) {
      unsigned long long* o = stamps + (size_t)gridDim.x * 8 + (size_t)gridDim.x * 128 + ((size_t)blockIdx.x * 8 + wave) * 16;
      for (int k = 0; k < 12; ++k) o[k] = ep[k];
    }
  }
#undef EP_MARK
#undef EP_ADD
#undef RR_SETUP_SRC
#undef RR_DMA
#undef RR_BLK
#undef RR_SBAR
}



unsigned long long* g_stamps = nullptr;   // diagnostic only (rr_set_gemm_stamps)
std::atomic<int> g_resid_touch{0};       // rr_set_tuning("resid_touch"): L2 touch of the next pass's residual rows; off since the rows themselves are requested a pass ahead (r03: 100.8 -> 99.8 ms)
std::atomic<int> g_variant{-1};          // tuning override (rr_set_gemm_variant / RR_GEMM_VARIANT); -1: shape heuristic

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the function: remember per device ordinal where
// it has been set (several rr_handles on different devices may live in one process).  Devices >= 64: set every launch.
inline hipError_t ensure_lds_attr(const void* kern, int lds_bytes, std::atomic<unsigned long long>& mask) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release);
  return hipSuccess;
}
// CUs of the current device (rounded down to a multiple of 8), cached per device ordinal
inline int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c) return c; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  const int n = prop.multiProcessorCount & ~7;
  if (dev >= 0 && dev < 64) cache[dev].store(n, std::memory_order_relaxed);
  return n;
}
// diagnostic (rr_set_tuning "gemm_grid_cus"): the persistent kernels run on this many CUs only (a multiple of 8; 0 = all) — what an
// epilogue costs when fewer CUs share the memory system (tools/gemm_epilogue_timeline.py --tuning gemm_grid_cus=128)
std::atomic<int> g_grid_cus{0};
inline int persistent_cus() {
  const int n = device_cus(), lim = g_grid_cus.load();
  return lim >= 8 && lim < n ? (lim & ~7) : n;
}

int g_stagger = 0;                        // start-skew unit in s_sleep(127) steps (rr_set_gemm_stagger)
std::atomic<int> g_resid_fast{1};         // rr_set_tuning("resid_fast"): plain fp32 residual GEMMs on the split forms' epilogue (SPLIT 4)
std::atomic<int> g_desync{0};             // rr_set_tuning("gemm_desync"): percent of the modelled tile period the eight XCDs are spread over
// start skew of the persistent kernel, as the kernel's stagger_unit argument (100 + sleeps of 256 cycles per XCD index)
inline int desync_arg(int Kd, int epilogue, int nwg, int n_cu) {
  const int pct = g_desync.load();
  if (pct <= 0 || nwg < 8 * n_cu) return 0;                              // fewer than 8 tiles per CU: the skew would cost more than it spreads
  const long main_cyc = (long)(Kd / 64) * 2700;                          // measured main loop, cycles per 256x256x64 K-tile
  const long epi_cyc = epilogue == EPI_BIAS_RESID_F32 ? 30000 : epilogue == EPI_BIAS_GELU_BF16 ? 16000 : 8000;
  const long u = (main_cyc + epi_cyc) * pct / 100 / 8 / 256;
  return u > 0 ? 100 + (int)(u > 1000 ? 1000 : u) : 0;
}
int g_persistent = 1;                     // rr_set_tuning("persistent_gemm"): 1 = variant 14 for large problems, 0 = variant 12

// persistent variant: one workgroup per CU (160 KiB of LDS each), grid = number of CUs rounded down to a multiple of 8
template <int DT>
hipError_t launch_hp(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                     int ldr, void* C, int ldc, int M, int N, int Kd, int epilogue, hipStream_t st, LnResid ln) {
  if (N & 7) return hipErrorInvalidValue;
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, nwg = tiles_m * tiles_n;
  const int n_cu = persistent_cus();
  if (n_cu < 8) return hipErrorInvalidValue;
  constexpr int lds_bytes = 160 * 1024;
  dim3 grid(nwg < n_cu ? ((nwg + 7) & ~7) : n_cu), block(512);
  unsigned long long* stamps = g_stamps;
  const int desync = desync_arg(Kd, epilogue, nwg, n_cu);
  const int mrev = nwg >= 2 * n_cu ? rr_m_direction_next() : 0;      // large launches only take part in the alternation
#define RR_GEMM_CASE_F(E, F)                                                                                  \
  {                                                                                                           \
    auto kern = gemm_kernel_hp<E, DT, 0, 0, F>;                                                               \
    static std::atomic<unsigned long long> attr_mask{0};     /* one bit per device ordinal: the attribute is per device */ \
    {                                                                                                         \
      hipError_t e = ensure_lds_attr((const void*)kern, lds_bytes, attr_mask);                                \
      if (e != hipSuccess) return e;                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd,  \
                       tiles_n, nwg, stamps, ln, (((g_stagger >= 50 && g_stagger <= 57) || g_stagger == 59) ? g_stagger : desync) + 1000 * mrev);       \
  }
#define RR_GEMM_CASE(E)                                                                                       \
  case E: {                                                                                                   \
    if (ln.in_stats) return hipErrorInvalidValue;            /* folded LayerNorm: the three forms below only */ \
    RR_GEMM_CASE_F(E, false)                                                                                  \
    break;                                                                                                    \
  }
#define RR_GEMM_CASE_FOLD(E)                                                                                  \
  case E: {                                                                                                   \
    if (ln.in_stats) RR_GEMM_CASE_F(E, true)                                                                  \
    else RR_GEMM_CASE_F(E, false)                                                                             \
    break;                                                                                                    \
  }
  const int split = (ln.r_hi ? 1 : 0) | (ln.lo_out ? 2 : 0) | (((ln.r_hi || ln.lo_out) && (ln.flags & 2)) ? 8 : 0);
  if (split && epilogue != EPI_BIAS_RESID_F32) return hipErrorInvalidValue;
#define RR_GEMM_SPLIT_CASE(S)                                                                                 \
  case S: {                                                                                                   \
    auto kern = gemm_kernel_hp<EPI_BIAS_RESID_F32, DT, S>;                                                    \
    static std::atomic<unsigned long long> attr_mask{0};                                                      \
    {                                                                                                         \
      hipError_t e = ensure_lds_attr((const void*)kern, lds_bytes, attr_mask);                                \
      if (e != hipSuccess) return e;                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd,  \
                       tiles_n, nwg, stamps, ln, (((g_stagger >= 50 && g_stagger <= 57) || g_stagger == 59) ? g_stagger : desync) + 1000 * mrev);       \
    return hipGetLastError();                                                                                 \
  }
  switch (split) {
    case 0: break;
    RR_GEMM_SPLIT_CASE(1)
    RR_GEMM_SPLIT_CASE(2)
    RR_GEMM_SPLIT_CASE(3)
    RR_GEMM_SPLIT_CASE(9)
    RR_GEMM_SPLIT_CASE(10)
    RR_GEMM_SPLIT_CASE(11)
  }
  // the plain fp32 residual stream (the fp8 configuration's attention-out, folding switched off) on the split forms' epilogue:
  // residual rows requested a pass ahead, 8-column chunks (rr_set_tuning "resid_fast", default 1; bit-identical to the older form)
  if (epilogue == EPI_BIAS_RESID_F32 && resid && !ln.x16 && !ln.in_stats && g_resid_fast.load() && !(N & 7)) {
    switch (4) { RR_GEMM_SPLIT_CASE(4) }
  }
#undef RR_GEMM_SPLIT_CASE
  switch (epilogue) {
    RR_GEMM_CASE_FOLD(EPI_BIAS_BF16)
    RR_GEMM_CASE_FOLD(EPI_BIAS_GELU_BF16)
    RR_GEMM_CASE_FOLD(EPI_BIAS_F32)
    RR_GEMM_CASE(EPI_BIAS_TANH_BF16)
    case EPI_BIAS_RESID_F32: {                               /* fp32 stream: with / without the folded-LayerNorm producer outputs */
      if (ln.in_stats) return hipErrorInvalidValue;
      if (ln.x16) RR_GEMM_CASE_F(EPI_BIAS_RESID_F32, true)
      else RR_GEMM_CASE_F(EPI_BIAS_RESID_F32, false)
      break;
    }
    RR_GEMM_CASE(EPI_BIAS_QGELU_BF16)
    default: return hipErrorInvalidValue;
  }
#undef RR_GEMM_CASE
#undef RR_GEMM_CASE_FOLD
#undef RR_GEMM_CASE_F
  return hipGetLastError();
}

// diagnostic build of the persistent kernel (variant 15, tools/bench_gemm.py --epilogue-timeline): the production forms only
template <int DT>
hipError_t launch_hp_diag(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                          int ldr, void* C, int ldc, int M, int N, int Kd, int epilogue, hipStream_t st, LnResid ln) {
  if (N & 7) return hipErrorInvalidValue;
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, nwg = tiles_m * tiles_n;
  const int n_cu = persistent_cus();
  if (n_cu < 8 || nwg < n_cu) return hipErrorInvalidValue;
  constexpr int lds_bytes = 160 * 1024;
  dim3 grid(n_cu), block(512);
  const int split = (ln.r_hi ? 1 : 0) | (ln.lo_out ? 2 : 0) | (((ln.r_hi || ln.lo_out) && (ln.flags & 2)) ? 8 : 0);
#define RR_DIAG_LAUNCH(E, S, F) RR_DIAG_LAUNCH_D(E, S, F, 1)
#define RR_DIAG_LAUNCH_D(E, S, F, D)                                                                                      \
  {                                                                                                                       \
    auto kern = gemm_kernel_hp<E, DT, S, D, F>;                                                                            \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);         \
    if (e != hipSuccess) return e;                                                                                        \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, tiles_n, nwg, \
                       g_stamps, ln, (g_stagger >= 62 && g_stagger <= 64) ? g_stagger : desync_arg(Kd, epilogue, nwg, n_cu));       \
    return hipGetLastError();                                                                                             \
  }
  if (g_stagger >= 62 && g_stagger <= 64) {      // duplicated fragment reads (64: the same build without copies): QKV and FFN-down forms only
    if (epilogue == EPI_BIAS_RESID_F32 && split == 3) RR_DIAG_LAUNCH_D(EPI_BIAS_RESID_F32, 3, false, 3)
    if (epilogue == EPI_BIAS_BF16 && split == 0 && ln.in_stats) RR_DIAG_LAUNCH_D(EPI_BIAS_BF16, 0, true, 3)
    return hipErrorInvalidValue;
  }
  if (epilogue == EPI_BIAS_RESID_F32 && split == 3) RR_DIAG_LAUNCH(EPI_BIAS_RESID_F32, 3, false)
  if (epilogue == EPI_BIAS_RESID_F32 && split == 11) RR_DIAG_LAUNCH(EPI_BIAS_RESID_F32, 11, false)
  if (epilogue == EPI_BIAS_RESID_F32 && split == 0 && !ln.x16) RR_DIAG_LAUNCH(EPI_BIAS_RESID_F32, 0, false)
  if (epilogue == EPI_BIAS_BF16 && split == 0 && ln.in_stats) RR_DIAG_LAUNCH(EPI_BIAS_BF16, 0, true)
  if (epilogue == EPI_BIAS_GELU_BF16 && split == 0 && ln.in_stats) RR_DIAG_LAUNCH(EPI_BIAS_GELU_BF16, 0, true)
#undef RR_DIAG_LAUNCH
#undef RR_DIAG_LAUNCH_D
  return hipErrorInvalidValue;
}

template <bool LDS_EPI, int DT>
hipError_t launch_h(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                    int ldr, void* C, int ldc, int M, int N, int Kd, int epilogue, hipStream_t st, LnResid ln) {
  if (LDS_EPI && (N & 7)) return hipErrorInvalidValue;
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, nwg = tiles_m * tiles_n;
  constexpr int ring = 8 * 128 * 128, stage_f32 = 128 * (256 * 4 + 16), stage_b16 = 256 * (256 * 2 + 16);
  constexpr int so = LDS_EPI ? (stage_f32 > stage_b16 ? stage_f32 : stage_b16) : 0;
  constexpr int lds_bytes = ring > so ? ring : so;
  dim3 grid(nwg), block(512);
  unsigned long long* stamps = g_stamps;
#define RR_GEMM_CASE(E)                                                                                       \
  case E: {                                                                                                   \
    auto kern = gemm_kernel_h<E, LDS_EPI, DT>;                                                                \
    static std::atomic<unsigned long long> attr_mask{0};     /* one bit per device ordinal: the attribute is per device */ \
    {                                                                                                         \
      hipError_t e = ensure_lds_attr((const void*)kern, lds_bytes, attr_mask);                                \
      if (e != hipSuccess) return e;                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd,  \
                       tiles_n, nwg, stamps, ln, nwg > 512 ? g_stagger : 0);                                                             \
    break;                                                                                                    \
  }
  if (g_variant.load() == 13) {   // diagnostic timeline build (tools/bench_gemm.py --timeline): bf16, bias -> 16-bit only
    if (!LDS_EPI || DT != 0 || epilogue != EPI_BIAS_BF16) return hipErrorInvalidValue;
    auto kern = gemm_kernel_h<EPI_BIAS_BF16, true, 0, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, tiles_n, nwg,
                       stamps, ln, g_stagger);
    return hipGetLastError();
  }
  switch (epilogue) {
    RR_GEMM_CASE(EPI_BIAS_BF16)
    RR_GEMM_CASE(EPI_BIAS_GELU_BF16)
    RR_GEMM_CASE(EPI_BIAS_F32)
    RR_GEMM_CASE(EPI_BIAS_TANH_BF16)
    RR_GEMM_CASE(EPI_BIAS_RESID_F32)
    RR_GEMM_CASE(EPI_BIAS_QGELU_BF16)
    default: return hipErrorInvalidValue;
  }
#undef RR_GEMM_CASE
  return hipGetLastError();
}


template <int BM, int BN, int WM, int WN, int STAGES, bool LDS_EPI = false, int DT = 0>
hipError_t launch_cfg(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                      int ldr, void* C, int ldc, int M, int N, int Kd, int epilogue, hipStream_t st, LnResid ln) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
  constexpr int ring_bytes = STAGES * (BM + BN) * BK * 2;
  // staging image: f32 -> one row group (BM/WM rows x (4 BN + 16) B); bf16 -> the whole tile (BM x (2 BN + 16) B)
  constexpr int stage_f32 = (BM / WM) * (BN * 4 + 16), stage_b16 = BM * (BN * 2 + 16);
  constexpr int stage_out_bytes = LDS_EPI ? (stage_f32 > stage_b16 ? stage_f32 : stage_b16) : 0;
  constexpr int lds_bytes = ring_bytes > stage_out_bytes ? ring_bytes : stage_out_bytes;
  static_assert(lds_bytes <= 160 * 1024, "LDS budget");
  if (LDS_EPI && (N & 7)) return hipErrorInvalidValue;
  dim3 grid(nwg), block(WM * WN * 64);
  unsigned long long* stamps = g_stamps;
#define RR_GEMM_CASE(E)                                                                                       \
  case E: {                                                                                                   \
    auto kern = gemm_kernel_s<BM, BN, WM, WN, STAGES, E, LDS_EPI, DT>;                                        \
    static std::atomic<unsigned long long> attr_mask{0};     /* one bit per device ordinal: the attribute is per device */ \
    {                                                                                                         \
      hipError_t e = ensure_lds_attr((const void*)kern, lds_bytes, attr_mask);                                \
      if (e != hipSuccess) return e;                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd,  \
                       tiles_n, nwg, stamps, (BM == 256 && BN == 256 && nwg > 512) ? g_stagger : 0, ln);     \
    break;                                                                                                    \
  }
  switch (epilogue) {
    RR_GEMM_CASE(EPI_BIAS_BF16)
    RR_GEMM_CASE(EPI_BIAS_GELU_BF16)
    RR_GEMM_CASE(EPI_BIAS_F32)
    RR_GEMM_CASE(EPI_BIAS_TANH_BF16)
    RR_GEMM_CASE(EPI_BIAS_RESID_F32)
    RR_GEMM_CASE(EPI_BIAS_QGELU_BF16)
    default: return hipErrorInvalidValue;
  }
#undef RR_GEMM_CASE
  return hipGetLastError();
}


}  // namespace

// tuning hook (tools/bench_gemm.py): -1 = shape heuristic
extern "C" int rr_set_gemm_variant(int v) {
  if (v < -1 || v > 15) return -1;
  g_variant.store(v);
  return 0;
}
extern "C" int rr_set_resid_touch(int on) {
  g_resid_touch.store(on != 0);
  return 0;
}
extern "C" int rr_set_gemm_persistent(int on) {
  g_persistent = on != 0;
  return 0;
}
extern "C" int rr_set_gemm_desync(int pct) {
  if (pct < 0 || pct > 400) return -1;
  g_desync.store(pct);
  return 0;
}
extern "C" int rr_set_resid_fast(int on) { g_resid_fast.store(on != 0); return 0; }
extern "C" int rr_set_gemm_stagger(int unit) {
  if (unit < 0 || unit > 64) return -1;
  g_stagger = unit;
  return 0;
}
// diagnostic: DEVICE buffer of 4 x uint64 per workgroup receiving cycle stamps, or NULL to switch off
extern "C" int rr_set_gemm_stamps(void* device_buf) {
  g_stamps = (unsigned long long*)device_buf;
  return 0;
}

std::atomic<int> g_m_alternate{1}, g_m_counter{0};       // rr_set_tuning("m_alternate"), default on: see rr_m_direction_next (rr_common.h)
extern "C" int rr_set_m_alternate(int on) { g_m_alternate.store(on != 0); g_m_counter.store(0); return 0; }
int rr_m_direction_next() { return g_m_alternate.load() ? (g_m_counter.fetch_add(1) & 1) : 0; }
std::atomic<int> g_resid_split{1};       // rr_set_tuning("resid_split")
// Smallest problem (in 256 x 256 tiles) that runs the persistent ring; below it the 128 x 128 two-stage kernel.  Measured per
// shape at the strong-scaling shard sizes of one K = 100 query (profiles/r04_e_midsize_variants.log, M = 6 656 / 12 800 / 25 600):
// the ring wins from ~150 tiles on (one tile per CU on 60 % of the chip beats three 128 x 128 tiles per CU: QKV 26 vs 37 us at
// 234 tiles, attention-out 33 vs 43 us at 150) and loses at 78 (28 vs 24 us).  Rounds 1-3 used 512.
std::atomic<int> g_ring_min_tiles{128};  // rr_set_tuning("gemm_ring_min_tiles")
std::atomic<int> g_small_half_rows{1};   // rr_set_tuning("gemm_small_half_rows"): 64 x 128 tiles below two 128 x 128 workgroups per CU
extern "C" int rr_set_gemm_small_half_rows(int on) { g_small_half_rows.store(on != 0); return 0; }
extern "C" int rr_set_gemm_grid_cus(int n) { g_grid_cus.store(n < 0 ? 0 : n); return 0; }
extern "C" int rr_set_gemm_ring_min_tiles(int n) { if (n < 1) return -1; g_ring_min_tiles.store(n); return 0; }
extern "C" int rr_set_resid_split(int on) { g_resid_split.store(on != 0); return 0; }
extern "C" int rr_get_resid_split(void) { return g_resid_split.load(); }
// The shape heuristic of rr_launch_gemm_fold, for callers that must know beforehand whether the split residual stream
// is available (every residual GEMM of a stack has the same M x N, so the answer holds for producer and consumer alike).
bool rr_gemm_split_ok(int M, int N) {
  static const bool env_variant = getenv("RR_GEMM_VARIANT") != nullptr;     // read once: an environment override pins a kernel
  if ((g_variant.load() >= 0 && g_variant.load() != 15) || !g_persistent || env_variant) return false;   // ("resid_split" on / off is the caller's: a handle option)
  return (long)((M + 255) / 256) * ((N + 255) / 256) >= g_ring_min_tiles.load() && !(N & 7);
}

hipError_t rr_launch_gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias,
                          const float* resid, int ldr, void* C, int ldc, int M, int N, int Kd,
                          int epilogue, int dt, hipStream_t st) {
  return rr_launch_gemm_fold(A, lda, W, ldw, bias, resid, ldr, nullptr, nullptr, nullptr, GemmFold{}, C, ldc, M, N, Kd, epilogue, dt, st);
}

hipError_t rr_launch_gemm_ln(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias,
                             const float* resid, int ldr, const float* ln_stats, const float* ln_gamma,
                             const float* ln_beta, void* C, int ldc, int M, int N, int Kd, int epilogue, int dt,
                             hipStream_t st) {
  return rr_launch_gemm_fold(A, lda, W, ldw, bias, resid, ldr, ln_stats, ln_gamma, ln_beta, GemmFold{}, C, ldc, M, N, Kd, epilogue, dt, st);
}

hipError_t rr_launch_gemm_fold(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias,
                               const float* resid, int ldr, const float* ln_stats, const float* ln_gamma,
                               const float* ln_beta, const GemmFold& fold, void* C, int ldc, int M, int N, int Kd,
                               int epilogue, int dt, hipStream_t st) {
  if (dt != 0 && dt != 1) return hipErrorInvalidValue;
  if (ln_stats && (epilogue != EPI_BIAS_RESID_F32 || !ln_gamma || !ln_beta)) return hipErrorInvalidValue;
  // folded LayerNorm: producer outputs belong to the fp32 residual epilogue; the consumer form needs its column sums
  if (fold.x16 && (epilogue != EPI_BIAS_RESID_F32 || !fold.part || fold.nparts != (N + 127) / 128 || (fold.ldx & 3) || (N & 7)))
    return hipErrorInvalidValue;
  if ((fold.in_stats != nullptr) != (fold.csum != nullptr)) return hipErrorInvalidValue;
  if (fold.in_stats && (epilogue == EPI_BIAS_RESID_F32)) return hipErrorInvalidValue;
  const LnResid ln{(const float2*)ln_stats, ln_gamma, ln_beta, fold.x16, fold.ldx, (float2*)fold.part, fold.nparts,
                   (const float2*)fold.in_stats, fold.csum, (g_resid_touch.load() & 1) | (fold.lo_bits == 8 ? 2 : 0), fold.r_hi, fold.r_lo, fold.ld16, fold.lo_out};
  const bool split = fold.r_hi || fold.r_lo || fold.lo_out;
  if (split) {
    if (epilogue != EPI_BIAS_RESID_F32 || (fold.r_hi == nullptr) != (fold.r_lo == nullptr) || (fold.ld16 & 3)) return hipErrorInvalidValue;
    if (!fold.x16 || fold.ldx != fold.ld16 || (fold.ldx & 7) || (N & 7)) return hipErrorInvalidValue;   // the hi half is the x16 rows; 16-byte chunks of 8
    if (fold.lo_bits != 16 && fold.lo_bits != 8) return hipErrorInvalidValue;
  }
  if (M <= 0 || N <= 0 || Kd <= 0) return hipErrorInvalidValue;
  if (Kd % BK != 0 || (lda & 7) || (ldw & 7) || (N & 3) || (ldc & 3)) return hipErrorInvalidValue;
  if (epilogue == EPI_BIAS_RESID_F32 && !fold.r_hi && (!resid || (ldr & 3))) return hipErrorInvalidValue;
  static std::once_flag env_once;
  std::call_once(env_once, [] {
    const char* e = getenv("RR_GEMM_VARIANT");
    if (e && g_variant.load() < 0) g_variant.store(atoi(e));
  });
  int v = g_variant.load();
  if (v < 0) {
    // big problems: 256x256 tiles, persistent half-tile LDS ring (variant HP) with the LDS-staged coalesced epilogue; small ones:
    // 128x128 so the grid still fills 256 CUs (measured with tools/bench_gemm.py --stamps, profiles/).
    const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    v = tiles256 >= g_ring_min_tiles.load() ? ((N & 7) ? 11 : (g_persistent ? 14 : 12)) : 0;   // 14: persistent ring (one workgroup per CU walks its tiles)
  }
  // Below the ring: 64 x 128 tiles where 128 x 128 ones leave the chip less than two workgroups per CU (the strong-scaling shard
  // shapes of one K = 100 query: 13 pairs = 312 tiles of 128 x 128 on 256 CUs, i.e. 56 CUs with two tiles and 200 with one; halved
  // tiles balance 624 over the chip — attention-out 24 -> 21 us, FFN-down 51 -> 49 us, profiles/r04 "64 x 128").
  // A row's values do not depend on the tile shape (same K walk per column, same accumulation order), so this moves time only.
  bool half_rows = false;
  if (v == 0) {
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    half_rows = g_small_half_rows.load() && tiles128 < 2L * (device_cus() > 0 ? device_cus() : 256);
  }
  if (fold.x16) {   // the producer side of the folded LayerNorm lives in the LDS-staged epilogues only
    if (v == 0) v = 20;
    else if (v != 10 && v != 12 && v != 14 && v != 15 && v != 20) return hipErrorInvalidValue;
  }
  if (split && v != 14 && v != 15) return hipErrorInvalidValue;     // the split residual stream lives in the persistent ring kernel only
  if (dt == 1) {   // fp16 operands: the production configurations only
    switch (v) {
      case 0:
        if (half_rows) return launch_cfg<64, 128, 2, 2, 2, false, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
        return launch_cfg<128, 128, 2, 2, 2, false, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 20:
        if (half_rows) return launch_cfg<64, 128, 2, 2, 2, true, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
        return launch_cfg<128, 128, 2, 2, 2, true, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 2: return launch_cfg<256, 256, 2, 4, 2, false, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 10: return launch_cfg<256, 256, 2, 4, 2, true, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 11: return launch_h<false, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 12: return launch_h<true, 1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 14: return launch_hp<1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      case 15: return launch_hp_diag<1>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      default: return hipErrorInvalidValue;
    }
  }
#define RR_CFG(BM_, BN_, WM_, WN_, ST_) \
  return launch_cfg<BM_, BN_, WM_, WN_, ST_>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln)
  switch (v) {
    case 0:
      if (half_rows) RR_CFG(64, 128, 2, 2, 2);
      RR_CFG(128, 128, 2, 2, 2);
    case 20:
      if (half_rows) return launch_cfg<64, 128, 2, 2, 2, true>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
      return launch_cfg<128, 128, 2, 2, 2, true>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
    case 1: RR_CFG(128, 128, 2, 2, 4);
    case 2: RR_CFG(256, 256, 2, 4, 2);
    case 3: RR_CFG(256, 128, 4, 2, 3);
    case 10: return launch_cfg<256, 256, 2, 4, 2, true>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
    case 11: return launch_h<false, 0>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
    case 12:
    case 13: return launch_h<true, 0>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
    case 14: return launch_hp<0>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st, ln);
    default: return hipErrorInvalidValue;
  }
#undef RR_CFG
}
