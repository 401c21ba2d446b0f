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
//   * STAGES-deep ring of K-tiles, one raw s_barrier per K-tile, counted s_waitcnt vmcnt(N) that leaves
//     the STAGES-2 youngest tiles in flight across the barrier (never a drain in the steady state);
//   * the MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends up with 4 consecutive output columns of one output row: bias/GELU/residual epilogues
//     work on float4 and stores are 8/16-byte vectors;
//   * blockIdx -> tile map is XCD-aware (bijective remap, T1): the workgroups that share an
//     activation row-panel run on one XCD so the panel is fetched from HBM once per XCD L2.
#include "rr_common.h"

#include <cstdlib>

namespace {

constexpr int BK = 64;

// LDS-DMA of 64 x 16 B: LDS destination = wave-uniform byte address (M0) + lane*16, per-lane global source.
// Issued from inline asm on purpose: hipcc's waitcnt pass treats the builtin form like a FLAT access and
// then degrades every later `s_waitcnt lgkmcnt(N)` in the loop to lgkmcnt(0), which serialises the
// ds_read -> MFMA software pipeline.  The DMA is counted by hand (wait_vmcnt below); no compiler-visible
// VMEM load lives inside the main loop (cdna_hip_programming.md §5.7 item 1).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst_wave_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_wave_base)
      : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }

// ---- diagnostic cycle stamps (tools/bench_gemm.py --stamps): block entry / first tile ready / main loop done /
// epilogue done, written by lane 0 of wave 0 into a buffer no kernel reads.  nullptr in every product launch.
__device__ __forceinline__ void stamp(unsigned long long* stamps, int slot) {
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
}

// Variant P ("pipelined"): 32x32x16 MFMA, two register fragment sets ping-ponged per 16-deep k-step, the tile
// barrier in the middle of the MFMA stream.
template <int BM, int BN, int WM, int WN, int STAGES, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel_p(const bf16_t* __restrict__ A, int lda,
                                                             const bf16_t* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, int ldr,
                                                             void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                             int tiles_n, int nwg, unsigned long long* stamps) {
  stamp(stamps, 0);
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;            // per-wave output tile
  constexpr int MT = TM / 32, NT = TN / 32;            // 32x32 MFMA tiles per wave
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW;    // 1-KiB DMA pieces (8 rows x 128 B) per wave per tile
  constexpr int PIECES = PA + PW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into whole pieces per wave");
  static_assert(TM % 32 == 0 && TN % 32 == 0, "per-wave tile must be a multiple of the 32x32 MFMA");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char lds[];   // STAGES * STAGE_BYTES, the only LDS object

  // ---- XCD-aware bijective block remap: blocks b, b+8, ... share an XCD (round-robin dispatch);
  // give each XCD a contiguous run of tiles.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;   // n fastest: A panel reused from L2
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- staging sources: lane -> (row, 16-byte chunk) of each of this wave's pieces
  const bf16_t* a_src[PA];
  const bf16_t* w_src[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave * PA + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);              // source chunk that lands in slot lane&7
    a_src[i] = A + (size_t)min(m0 + r, M - 1) * lda + c * 8;   // clamp: padded rows re-read a valid row
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = (wave * PW + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w_src[i] = W + (size_t)min(n0 + r, N - 1) * ldw + c * 8;
  }
  const uint32_t lds_base = lds_addr(lds);
  auto stage = [&](int buf, int k0) {
    const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + wave * PA * 1024);
    const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + A_BYTES + wave * PW * 1024);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(a_src[i] + k0, a_dst + i * 1024);
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16(w_src[i] + k0, w_dst + i * 1024);
  };

  // ---- fragment addressing: lane reads row (lane&31) of a 32-row block, chunk 2*s + (lane>>5) of k16-step s.
  // The swizzle term (row>>1)&7 does not depend on the 32-row block index, so block mt is a +4096 immediate.
  const int r32 = lane & 31, hh = lane >> 5;
  const int a_row = wm * TM + r32, w_row = wn * TN + r32;
  const int a_x = (a_row >> 1) & 7, w_x = (w_row >> 1) & 7;
  auto load_frags = [&](const char* tile_base, int s, bf16x8 (&af)[MT], bf16x8 (&wf)[NT]) {
    const char* ab = tile_base + a_row * 128 + (((2 * s + hh) ^ a_x) << 4);
    const char* wb = tile_base + A_BYTES + w_row * 128 + (((2 * s + hh) ^ w_x) << 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[t] = *(const bf16x8*)(wb + t * 4096);
#pragma unroll
    for (int t = 0; t < MT; ++t) af[t] = *(const bf16x8*)(ab + t * 4096);
  };

  f32x16 acc[NT][MT];   // lane holds m = mt*32 + (lane&31), n = nt*32 + (r&3) + 8(r>>2) + 4(lane>>5)
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = Kd / BK;
#pragma unroll
  for (int s = 0; s < STAGES; ++s)
    if (s < nk) stage(s, s * BK);

  bf16x8 afA[MT], wfA[NT], afB[MT], wfB[NT];   // two register fragment sets (k16-step ping-pong)
  wait_tiles<PIECES, STAGES - 1>(min(nk, STAGES) - 1);   // tile 0 landed
  __builtin_amdgcn_s_barrier();
  stamp(stamps, 1);
  load_frags(lds, 0, afA, wfA);

#define RR_MFMA_BLOCK(AF, WF)                                                                        \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) \
      acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[nt], AF[mt], acc[nt][mt], 0, 0, 0);

  // The sched_barriers pin "issue the next step's 6 LDS reads, THEN this step's MFMAs": the reads' latency is
  // covered by a whole MFMA block instead of being waited for right before use.
#define RR_SB() __builtin_amdgcn_sched_barrier(0)
#define RR_STEPS_0_TO_2(tb)        \
  load_frags(tb, 1, afB, wfB);     \
  RR_SB();                         \
  RR_MFMA_BLOCK(afA, wfA)          \
  RR_SB();                         \
  load_frags(tb, 2, afA, wfA);     \
  RR_SB();                         \
  RR_MFMA_BLOCK(afB, wfB)          \
  RR_SB();                         \
  load_frags(tb, 3, afB, wfB);     \
  RR_SB();                         \
  RR_MFMA_BLOCK(afA, wfA)          \
  RR_SB();
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const char* tb = lds + (kt % STAGES) * STAGE_BYTES;
    RR_STEPS_0_TO_2(tb)
    // tile kt+1 must have landed; tiles kt+2 .. min(nk-1, kt+STAGES-1) may stay in flight
    wait_tiles<PIECES, STAGES - 2>(min(nk - 1, kt + STAGES - 1) - (kt + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of tile kt are complete ...
    __builtin_amdgcn_s_barrier();                        // ... and everyone's: its buffer may be refilled
    if (kt + STAGES < nk) stage(kt % STAGES, (kt + STAGES) * BK);
    load_frags(lds + ((kt + 1) % STAGES) * STAGE_BYTES, 0, afA, wfA);
    RR_SB();
    RR_MFMA_BLOCK(afB, wfB)
    RR_SB();
  }
  {   // last K-tile: nothing left to prefetch
    const char* tb = lds + ((nk - 1) % STAGES) * STAGE_BYTES;
    RR_STEPS_0_TO_2(tb)
    RR_MFMA_BLOCK(afB, wfB)
  }
#undef RR_STEPS_0_TO_2
  stamp(stamps, 2);
#undef RR_SB
#undef RR_MFMA_BLOCK

  // ---- epilogue: per 32x32 tile a lane owns row m and 4 groups of 4 consecutive columns
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int gn = n0 + wn * TN + nt * 32 + 8 * g + 4 * hh;
      if (gn >= N) continue;
      const float4 bv = bias ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int gm = m0 + wm * TM + mt * 32 + r32;
        if (gm >= M) continue;
        float v0 = acc[nt][mt][4 * g + 0] + bv.x, v1 = acc[nt][mt][4 * g + 1] + bv.y,
              v2 = acc[nt][mt][4 * g + 2] + bv.z, v3 = acc[nt][mt][4 * g + 3] + bv.w;
        if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
        if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3); }
        if (EPI == EPI_BIAS_RESID_F32) {
          const float4 rv = *(const float4*)(resid + (size_t)gm * ldr + gn);
          v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
        }
        if (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RESID_F32) {
          *(float4*)((float*)Cv + (size_t)gm * ldc + gn) = make_float4(v0, v1, v2, v3);
        } else {
          *(uint2*)((bf16_t*)Cv + (size_t)gm * ldc + gn) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        }
      }
    }
  }
  stamp(stamps, 3);
}

// Variant S ("simple"): 16x16x32 MFMA, tile barrier at the top of each K-tile, fragment reads scheduled by the
// compiler inside the tile.
template <int BM, int BN, int WM, int WN, int STAGES, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel_s(const bf16_t* __restrict__ A, int lda,
                                                             const bf16_t* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, int ldr,
                                                             void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                             int tiles_n, int nwg, unsigned long long* stamps) {
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

  const bf16_t* a_src[PA];
  const bf16_t* w_src[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave * PA + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    a_src[i] = A + (size_t)min(m0 + r, M - 1) * lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = (wave * PW + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w_src[i] = W + (size_t)min(n0 + r, N - 1) * ldw + c * 8;
  }
  const uint32_t lds_base = lds_addr(lds);
  auto stage = [&](int buf, int k0) {
    const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + wave * PA * 1024);
    const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + A_BYTES + wave * PW * 1024);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(a_src[i] + k0, a_dst + i * 1024);
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16(w_src[i] + k0, w_dst + i * 1024);
  };

  f32x4 acc[NT][MT];   // lane holds m = mt*16 + (lane&15), n = nt*16 + (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = Kd / BK;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s * BK);

  unsigned long long t_wait = 0, t_bar = 0;   // diagnostic accumulators (only when stamps != nullptr)
  for (int kt = 0; kt < nk; ++kt) {
    unsigned long long tA = 0, tB = 0;
    if (stamps) tA = __builtin_amdgcn_s_memtime();
    wait_tiles<PIECES, STAGES - 2>(min(nk - 1, kt + STAGES - 2) - kt);
    if (stamps) tB = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();          // everyone's pieces of tile kt landed; compute(kt-1) is done
    if (stamps) { const unsigned long long tC = __builtin_amdgcn_s_memtime(); t_wait += tB - tA; t_bar += tC - tB; }
    if (kt == 0) stamp(stamps, 1);
    if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * BK);
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
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    }
  }
  stamp(stamps, 2);
  if (stamps && threadIdx.x == 0) { stamps[(size_t)blockIdx.x * 8 + 4] = t_wait; stamps[(size_t)blockIdx.x * 8 + 5] = t_bar; }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int gn = n0 + wn * TN + nt * 16 + (lane >> 4) * 4;
    if (gn >= N) continue;
    float4 bv = bias ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gm = m0 + wm * TM + mt * 16 + (lane & 15);
      if (gm >= M) continue;
      float v0 = acc[nt][mt][0] + bv.x, v1 = acc[nt][mt][1] + bv.y, v2 = acc[nt][mt][2] + bv.z,
            v3 = acc[nt][mt][3] + bv.w;
      if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
      if (EPI == EPI_BIAS_TANH_BF16) { v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3); }
      if (EPI == EPI_BIAS_RESID_F32) {
        const float4 rv = *(const float4*)(resid + (size_t)gm * ldr + gn);
        v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
      }
      if (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RESID_F32) {
        *(float4*)((float*)Cv + (size_t)gm * ldc + gn) = make_float4(v0, v1, v2, v3);
      } else {
        *(uint2*)((bf16_t*)Cv + (size_t)gm * ldc + gn) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
      }
    }
  }
  stamp(stamps, 3);
}

unsigned long long* g_stamps = nullptr;   // diagnostic only (rr_set_gemm_stamps)

template <int BM, int BN, int WM, int WN, int STAGES, bool PIPE>
hipError_t launch_cfg(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                      int ldr, void* C, int ldc, int M, int N, int Kd, int epilogue, hipStream_t st) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
  constexpr int lds_bytes = STAGES * (BM + BN) * BK * 2;
  dim3 grid(nwg), block(WM * WN * 64);
  unsigned long long* stamps = g_stamps;
#define RR_GEMM_CASE(E)                                                                                       \
  case E: {                                                                                                   \
    auto kern = PIPE ? gemm_kernel_p<BM, BN, WM, WN, STAGES, E> : gemm_kernel_s<BM, BN, WM, WN, STAGES, E>;                                                       \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      if (e != hipSuccess) return e;                                                                          \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd,  \
                       tiles_n, nwg, stamps);                                                                 \
    break;                                                                                                    \
  }
  switch (epilogue) {
    RR_GEMM_CASE(EPI_BIAS_BF16)
    RR_GEMM_CASE(EPI_BIAS_GELU_BF16)
    RR_GEMM_CASE(EPI_BIAS_F32)
    RR_GEMM_CASE(EPI_BIAS_TANH_BF16)
    RR_GEMM_CASE(EPI_BIAS_RESID_F32)
    default: return hipErrorInvalidValue;
  }
#undef RR_GEMM_CASE
  return hipGetLastError();
}

int g_variant = -1;   // tuning override: RR_GEMM_VARIANT=0..3 (unset: shape heuristic)

}  // namespace

// tuning hook (tools/bench_gemm.py): -1 = shape heuristic
extern "C" int rr_set_gemm_variant(int v) {
  if (v < -1 || v > 7) return -1;
  g_variant = v;
  return 0;
}
// diagnostic: DEVICE buffer of 4 x uint64 per workgroup receiving cycle stamps, or NULL to switch off
extern "C" int rr_set_gemm_stamps(void* device_buf) {
  g_stamps = (unsigned long long*)device_buf;
  return 0;
}

hipError_t rr_launch_gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias,
                          const float* resid, int ldr, void* C, int ldc, int M, int N, int Kd,
                          int epilogue, hipStream_t st) {
  if (M <= 0 || N <= 0 || Kd <= 0) return hipErrorInvalidValue;
  if (Kd % BK != 0 || (lda & 7) || (ldw & 7) || (N & 3) || (ldc & 3)) return hipErrorInvalidValue;
  if (epilogue == EPI_BIAS_RESID_F32 && (!resid || (ldr & 3))) return hipErrorInvalidValue;
  static bool env_read = false;
  if (!env_read) {
    const char* e = getenv("RR_GEMM_VARIANT");
    if (e && g_variant < 0) g_variant = atoi(e);
    env_read = true;
  }
  int v = g_variant;
  if (v < 0) {
    // big problems: 256x256 tiles (1 workgroup/CU, 8 waves); small ones: 128x128 so the grid still fills 256 CUs
    const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    v = tiles256 >= 512 ? 2 : 0;
  }
#define RR_CFG(BM_, BN_, WM_, WN_, ST_, P_) \
  return launch_cfg<BM_, BN_, WM_, WN_, ST_, P_>(A, lda, W, ldw, bias, resid, ldr, C, ldc, M, N, Kd, epilogue, st)
  switch (v) {
    case 0: RR_CFG(128, 128, 2, 2, 2, false);
    case 1: RR_CFG(128, 128, 2, 2, 4, false);
    case 2: RR_CFG(256, 256, 2, 4, 2, false);
    case 3: RR_CFG(256, 128, 4, 2, 3, false);
    case 4: RR_CFG(128, 128, 2, 2, 2, true);
    case 5: RR_CFG(128, 128, 2, 2, 3, true);
    case 6: RR_CFG(256, 256, 2, 4, 2, true);
    case 7: RR_CFG(256, 128, 4, 2, 3, true);
    default: return hipErrorInvalidValue;
  }
#undef RR_CFG
}
