// e4m3 (OCP fp8) GEMMs for gfx950 on the block-scaled matrix core:
//     C[M,N] = epi(sa[m] * sw[n] * (A8[M,K] . W8[N,K]^T) + bias[n])
// BASELINE configs[4] ("bert-large ... fp8 MFMA"): the QKV and FFN-up GEMMs of every BertLayer take
// e4m3 operands — activations quantised per ROW by the LayerNorm kernel that produces them (layernorm_q8_kernel: scale =
// row amax / 448), weights per OUTPUT CHANNEL at pack time — the GEMMs that feed the residual stream stay 16-bit
// (tests/tools/fp8_mix_study.py: e4m3 there reorders candidate lists).  On gfx950 the 2x rate only exists on the
// v_mfma_scale_f32_*_f8f6f4 instructions (the plain _fp8_fp8 forms run at the bf16 rate, MI355X_MICROARCH.md "Matrix
// cores"); block scales are fixed at 2^0 (E8M0 127) and the row / channel scales are applied to the fp32 accumulators.
//
// Two kernels:
//  * gemm_kernel_hp8 — the production kernel for >= 512 tiles: the persistent half-tile LDS ring of gemm_bf16.hip
//    (gemm_kernel_hp) with 128-BYTE K-tile rows holding 128 fp8 instead of 64 bf16, so ring slots, LDS-DMA pieces, swizzle,
//    barriers and counted vmcnt waits are byte for byte the same, on v_mfma_scale_f32_32x32x64_f8f6f4: one K-tile row is
//    two 64-deep k-steps exactly like the two 32-deep k-steps of the 16-bit tile, so the four-phase schedule (fragments
//    fetched one block ahead of their use) carries over 1:1 with the same register budget — a block is 2 MFMAs of 64
//    cycles instead of 8 of 16.  Operand lane map (tools/fp8_mfma32_probe.hip, exact integer data on the device): lane l
//    holds A[row l&31][k = 32 (l>>5) + j], j = 0..31 in byte order (= 16-byte chunks 2h, 2h+1 of the k-step's 64 bytes),
//    likewise B; C/D as every 32x32 MFMA (col = l&31, row = (reg&3) + 8 (reg>>2) + 4 (l>>5)).  Issued "swapped" (A-operand
//    = weight rows) so a lane ends up with 4 consecutive output columns per register group.
//  * gemm_kernel_f8 — small problems: 16x16x128 form, two-stage LDS-DMA, direct epilogue (lane map: tools/fp8_mfma_probe.hip).
#include "rr_common.h"

#include <atomic>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16_;
constexpr int BKB = 128;     // K-tile: 128 bytes = 128 fp8 per row

template <int N>
__device__ __forceinline__ void wait_vmcnt8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float gelu13(float x) { return gelu_erf_fast(x); }   // rr_common.h (the name is historical)

// EPI: 0 = 16bit(acc*s + b), 1 = 16bit(gelu(acc*s + b)), 2 = f32(acc*s + b);  s = scale * row_scale[m] * col_scale[n]
// (row_scale / col_scale may be null = 1)
template <int BM, int BN, int WM, int WN, int EPI, int DT>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel_f8(const uint8_t* __restrict__ A, int lda,
                                                              const uint8_t* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, float scale,
                                                              const float* __restrict__ row_scale,
                                                              const float* __restrict__ col_scale,
                                                              void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                              int tiles_n, int nwg) {
  constexpr int NW = WM * WN, STAGES = 2;
  constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
  constexpr int A_BYTES = BM * BKB, W_BYTES = BN * BKB, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW, PIECES = PA + PW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into whole pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  // XCD-aware: consecutive tiles of the list go to one XCD (same scheme as the 16-bit kernels)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  uint32_t a_off[PA], w_off[PW];      // byte offsets from the scalar base; chunk swizzle applied on the global side
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave * PA + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    a_off[i] = (uint32_t)((size_t)min(r, M - 1 - m0) * lda + c * 16);
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = (wave * PW + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w_off[i] = (uint32_t)((size_t)min(r, N - 1 - n0) * ldw + c * 16);
  }
  const uint8_t* const a_tile = A + (size_t)m0 * lda;    // 64-bit scalar tile origin + 32-bit in-tile offsets: any operand size
  const uint8_t* const w_tile = W + (size_t)n0 * ldw;
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

  const int nk = Kd / BKB;
  stage(0, 0);
  const int g2 = 2 * (lane >> 4);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt8<0>();                       // my pieces of tile kt (the only ones in flight) have landed
    __builtin_amdgcn_s_barrier();           // everyone's have; compute(kt-1) is done, its buffer is free
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BKB);
    const char* a_t = lds + (kt & 1) * STAGE_BYTES;
    const char* w_t = a_t + A_BYTES;
    i32x8 af[MT], wf[NT];
    auto frag = [&](const char* img, int row) {
      const i32x4 lo = *(const i32x4*)(img + swz128(row, g2));
      const i32x4 hi = *(const i32x4*)(img + swz128(row, g2 + 1));
      return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[t] = frag(w_t, wn * TN + t * 16 + (lane & 15));
#pragma unroll
    for (int t = 0; t < MT; ++t) af[t] = frag(a_t, wm * TM + t * 16 + (lane & 15));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[mt], acc[nt][mt], 0, 0, 0, 127, 0, 127);
  }

  // ---- direct epilogue: lane owns 4 consecutive n of one output row m
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int gn = n0 + wn * TN + nt * 16 + (lane >> 4) * 4;
    if (gn >= N) continue;
    const float4 bv = bias ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cw = col_scale ? *(const float4*)(col_scale + gn) : make_float4(1.f, 1.f, 1.f, 1.f);
    cw.x *= scale; cw.y *= scale; cw.z *= scale; cw.w *= scale;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gm = m0 + wm * TM + mt * 16 + (lane & 15);
      if (gm >= M) continue;
      const float rs = row_scale ? row_scale[gm] : 1.0f;
      float v0 = fmaf(acc[nt][mt][0], rs * cw.x, bv.x), v1 = fmaf(acc[nt][mt][1], rs * cw.y, bv.y),
            v2 = fmaf(acc[nt][mt][2], rs * cw.z, bv.z), v3 = fmaf(acc[nt][mt][3], rs * cw.w, bv.w);
      if (EPI == 1) { v0 = gelu13(v0); v1 = gelu13(v1); v2 = gelu13(v2); v3 = gelu13(v3); }
      if (EPI == 2) *(float4*)((float*)Cv + (size_t)gm * ldc + gn) = make_float4(v0, v1, v2, v3);
      else *(uint2*)((bf16_t*)Cv + (size_t)gm * ldc + gn) = make_uint2(pack2<DT>(v0, v1), pack2<DT>(v2, v3));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent half-tile ring on the 32x32x64 block-scaled MFMA (see the file header).  256x256 output tile, 8 waves
// (wr = wave>>2, wc = wave&3), wave tile 128x64 = 4 quadrants of 64x32; LDS ring of 8 half-tile slots of 16 KiB
// {A rows 0-127, B rows 0-127, B rows 128-255, A rows 128-255} x 2 tile parities, refilled by LDS-DMA as soon as a slot's
// fragments are in registers; per K-tile 4 phases (quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0)), each two blocks of
// 2 MFMAs (k-step 0, k-step 1), fragments read one block ahead; sync points X (after p1) and Y (after p3) with the same
// counted vmcnt literals as gemm_kernel_hp (2 pieces per half-tile per wave).  EPI 0: 16-bit out, 1: 16-bit erf-GELU,
// 3: erf-GELU -> e4m3 bytes (x.out_mul, a power of two, is the static scale of the whole tensor: the consumer multiplies its
// accumulators by 1 / out_mul), 4: fp32 out = acc * s + bias + residual row (optionally LayerNorm-recomputed from its
// statistics, as gemm_kernel_hp's fp32-stream residual epilogue does) — the FFN-down of the fp8 configuration.
struct F8Extra {
  float out_mul = 1.0f;             // EPI 3
  const float* resid = nullptr;     // EPI 4: residual rows [M, ldr] f32
  int ldr = 0;
  const float2* rstats = nullptr;   //        (mean, rstd) per row, or null: the rows are the residual themselves
  const float* rgamma = nullptr;
  const float* rbeta = nullptr;
};
template <int EPI, int DT>
__global__ __launch_bounds__(512) void gemm_kernel_hp8(const uint8_t* __restrict__ A, int lda,
                                                      const uint8_t* __restrict__ W, int ldw,
                                                      const float* __restrict__ bias, float scale,
                                                      const float* __restrict__ row_scale,
                                                      const float* __restrict__ col_scale,
                                                      void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                      int tiles_n, int nwg, F8Extra x) {
  bf16_t* const C = (bf16_t*)Cv;
  constexpr int BM = 256, BN = 256, HALF = 128 * 128;       // half-tile = 128 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int GROUP = Kd > 2048 ? 1 : 8;                      // L2-aware tile order, as gemm_kernel_hp (K bytes per row: Kd)

  const int bid = blockIdx.x, gstep = (int)(gridDim.x >> 3);
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, chunk_n = q8 + (xcd < r8 ? 1 : 0);
  int li = bid >> 3;
  if (li >= chunk_n) return;
  int m0, n0;
  const int tiles_m = nwg / tiles_n;
  auto tile_origin = [&](int gidx, int& m0_, int& n0_) {
    const int per_group = GROUP * tiles_n, grp = gidx / per_group, r = gidx - grp * per_group;
    const int rows = min(GROUP, tiles_m - grp * GROUP);
    const int tn = r / rows, tm = grp * GROUP + (r - tn * rows);
    m0_ = tm * BM;
    n0_ = tn * BN;
  };
  tile_origin(chunk0 + li, m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, h = lane >> 5;

  uint32_t so_a0[2], so_a1[2], so_b0[2], so_b1[2];
  const uint8_t *a_tile, *w_tile;
#define RR_SETUP_SRC(m0_, n0_)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
    const int r = (wave * 2 + i) * 8 + (lane >> 3);          /* row inside the half-tile */             \
    const int c = (lane & 7) ^ ((r >> 1) & 7);                                                          \
    so_a0[i] = (uint32_t)((size_t)min(r, M - 1 - (m0_)) * lda + c * 16);                                \
    so_a1[i] = (uint32_t)((size_t)min(128 + r, M - 1 - (m0_)) * lda + c * 16);                          \
    so_b0[i] = (uint32_t)((size_t)min(r, N - 1 - (n0_)) * ldw + c * 16);                                \
    so_b1[i] = (uint32_t)((size_t)min(128 + r, N - 1 - (n0_)) * ldw + c * 16);                          \
  }                                                                                                     \
  a_tile = A + (size_t)(m0_) * lda;                                                                     \
  w_tile = W + (size_t)(n0_) * ldw;
  RR_SETUP_SRC(m0, n0)
  const uint32_t lds_base = lds_addr(lds);
  const int nk = Kd / BKB, H = 4 * nk;
  // A K-tile index beyond the last one (the refills the last two K-tiles of an output tile would issue for K-tiles that do
  // not exist) is redirected: all lanes fetch the first 16 bytes of the last K-tile (one request, L1-resident) into a ring
  // slot nobody reads before the next output tile's prologue overwrites it.  Every K-tile therefore issues the same
  // DMA instructions: ONE loop body with literal vmcnt waits, no general-guard tail variant (whose register allocation
  // spilled accumulators and DMA offsets to scratch, and scratch traffic shares the vmcnt counter with the DMA).
#define RR_DMA(t_, J)                                                                                             \
  {                                                                                                               \
    const uint32_t dst_ = __builtin_amdgcn_readfirstlane(lds_base + (((t_) & 1) * 4 + (J)) * HALF + wave * 2048); \
    const bool real_ = (t_) < nk;                                                                                 \
    const size_t kb_ = (size_t)(real_ ? (t_) : nk - 1) * BKB;                                                     \
    const void* sb_ = ((J) == 0 || (J) == 3) ? (const void*)(a_tile + kb_) : (const void*)(w_tile + kb_);         \
    const uint32_t* so_ = (J) == 0 ? so_a0 : (J) == 3 ? so_a1 : (J) == 1 ? so_b0 : so_b1;                         \
    glds16_so(sb_, real_ ? so_[0] : 0u, dst_);                                                                    \
    glds16_so(sb_, real_ ? so_[1] : 0u, dst_ + 1024);                                                             \
  }

  // fragment of k-step ks for a 32-row block: the 32 bytes [64 ks + 32 h, +32) of the row = 16-byte chunks 4 ks + 2 h and
  // + 1 (the swizzle term is the same for every 32-row block: ((row >> 1) & 7) repeats every 16 rows)
  const int a_off = swz128(wr * 64 + (lane & 31), 2 * h);
  const int b_off = swz128(wc * 32 + (lane & 31), 2 * h);
  auto frag = [&](const char* p, int off) {
    const i32x4 lo = *(const i32x4*)(p + off);
    const i32x4 hi = *(const i32x4*)(p + (off ^ 16));
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto read_a = [&](const char* slot, int ks, i32x8 (&f)[2]) {
    const int o = ks ? (a_off ^ 64) : a_off;
    f[0] = frag(slot, o);
    f[1] = frag(slot + 32 * 128, o);
  };
  auto read_b = [&](const char* slot, int ks, i32x8& f) { f = frag(slot, ks ? (b_off ^ 64) : b_off); };

  f32x16 acc[4][2];     // [quadrant 2*hA+hB][mb]; lane: m = mb*32 + (lane&31), n = (reg&3) + 8 (reg>>2) + 4 h
  i32x8 AF0[2], AF1[2], B0K0, B0K1, B1K0, B1K1;

#define RR_BLK(Q, AF, BF)                                                                                           \
  _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                                  \
      acc[Q][mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(BF, AF[mb], acc[Q][mb], 0, 0, 0, 127, 0, 127);
#define RR_SBAR() __builtin_amdgcn_sched_barrier(0)
#define RR_PRIO(p) __builtin_amdgcn_s_setprio(p);
#define RR_SYNC(NLIT)                                    \
  {                                                      \
    RR_SBAR();                                           \
    wait_vmcnt8<NLIT>();                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    RR_SBAR();                                           \
  }
#define RR_KEEP6() asm volatile("" : "+v"(AF0[0]), "+v"(AF0[1]), "+v"(B0K0));

  // Per-row / per-column parameters of the tile (row scales, column scales, bias; EPI 4: gamma, beta) reach [152 KiB, 160 KiB)
  // by dword LDS-DMA pieces issued BEFORE the tile's first ring refill — as in gemm_kernel_hp: a global load in the epilogue
  // would queue behind the ten pieces of the next tile's prefetch in the wave's in-order vmcnt queue (80 KiB first); older
  // than every refill, the pieces only make the loop's counted waits stricter.  Absent vectors: the block holds 1 / 1 / 0.
  constexpr int PARAM_OFF = 152 * 1024;     // +0 row_scale[256], +1024 col_scale[256], +2048 bias[256], +3072 gamma, +4096 beta
  {
    float* const pb = (float*)(lds + PARAM_OFF);
    if (tid < 256) {
      if (!row_scale) pb[tid] = 1.0f;
      if (!col_scale) pb[256 + tid] = 1.0f;
      if (!bias) pb[512 + tid] = 0.0f;
    }
  }
  auto stage_params = [&](int m0_, int n0_) {
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));     // opaque: recomputed per tile, not hoisted out of the tile loop and spilled
    const int ln_ = lane_;
    if (wave < 4) {
      if (row_scale) glds4_so(row_scale + m0_, (uint32_t)(min(wave * 64 + ln_, M - 1 - m0_) * 4),
                              __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + wave * 256));
      if (col_scale) glds4_so(col_scale + n0_, (uint32_t)(min(wave * 64 + ln_, N - 1 - n0_) * 4),
                              __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 1024 + wave * 256));
      if (EPI == 4 && x.rstats) glds4_so(x.rgamma + n0_, (uint32_t)(min(wave * 64 + ln_, N - 1 - n0_) * 4),
                                         __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 3072 + wave * 256));
    } else {
      if (bias) glds4_so(bias + n0_, (uint32_t)(min((wave - 4) * 64 + ln_, N - 1 - n0_) * 4),
                         __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 2048 + (wave - 4) * 256));
      if (EPI == 4 && x.rstats) glds4_so(x.rbeta + n0_, (uint32_t)(min((wave - 4) * 64 + ln_, N - 1 - n0_) * 4),
                                         __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + 4096 + (wave - 4) * 256));
    }
  };
  stage_params(m0, n0);
  RR_DMA(0, 0) RR_DMA(0, 1) RR_DMA(0, 2) RR_DMA(0, 3)
  RR_DMA(1, 0) RR_DMA(1, 1) RR_DMA(1, 2)
  bool first_tile = true;
  for (;;) {                                                // one iteration per output tile of this workgroup
  if (first_tile) {
    wait_vmcnt8<4>();                                       // all of K-tile 0 (my pieces); leaves A0(1), B0(1) ... in flight
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][mb][r] = 0.f;
  read_a(lds + 0 * HALF, 0, AF0);
  read_b(lds + 1 * HALF, 0, B0K0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  RR_KEEP6()

#define RR_TILE()                                                                                    \
  {                                                                                                        \
    const char* sl = lds + (t & 1) * 4 * HALF;               /* this tile's slots: +0 A0, +1 B0, +2 B1, +3 A1 */ \
    const char* sn = lds + ((t + 1) & 1) * 4 * HALF;         /* next tile's */                              \
    /* ---- p0: quadrant (A0, B0) */                                                                        \
    read_a(sl + 0 * HALF, 1, AF1);                                                                         \
    read_b(sl + 1 * HALF, 1, B0K1);                                                                        \
    RR_SBAR();                                                                                             \
    RR_PRIO(3)                                                                                             \
    RR_BLK(0, AF0, B0K0)                                                                                   \
    RR_SBAR();                                                                                             \
    read_b(sl + 2 * HALF, 0, B1K0);                                                                        \
    RR_DMA(t + 1, 3)                                      /* A1 of the next tile (slot free since Y(t-1)) */ \
    RR_SBAR();                                                                                             \
    RR_PRIO(2)                                                                                             \
    RR_BLK(0, AF1, B0K1)                                                                                   \
    RR_SBAR();                                                                                             \
    /* ---- p1: quadrant (A0, B1) */                                                                        \
    read_b(sl + 2 * HALF, 1, B1K1);                                                                        \
    RR_SBAR();                                                                                             \
    RR_PRIO(1)                                                                                             \
    RR_BLK(1, AF0, B1K0)                                                                                   \
    RR_SBAR();                                                                                             \
    read_a(sl + 3 * HALF, 0, AF0);                                                                         \
    RR_SBAR();                                                                                             \
    RR_PRIO(0)                                                                                             \
    RR_BLK(1, AF1, B1K1)                                                                                   \
    RR_SYNC(4)   /* X: A0(t+1), B0(t+1) landed */       \
    /* ---- p2: quadrant (A1, B1) */                                                                        \
    read_a(sl + 3 * HALF, 1, AF1);                                                                         \
    RR_DMA(t + 2, 0)                                      /* slots A0, B0 (free since X) */               \
    RR_SBAR();                                                                                             \
    RR_PRIO(3)                                                                                             \
    RR_BLK(3, AF0, B1K0)                                                                                   \
    RR_SBAR();                                                                                             \
    RR_DMA(t + 2, 1)                                                                                    \
    RR_SBAR();                                                                                             \
    RR_PRIO(2)                                                                                             \
    RR_BLK(3, AF1, B1K1)                                                                                   \
    RR_SBAR();                                                                                             \
    /* ---- p3: quadrant (A1, B0) */                                                                        \
    RR_PRIO(1)                                                                                             \
    RR_BLK(2, AF0, B0K0)                                                                                   \
    RR_SBAR();                                                                                             \
    read_a(sn + 0 * HALF, 0, AF0);                           /* (stale bytes after the last K-tile: re-read at the next tile's start) */ \
    read_b(sn + 1 * HALF, 0, B0K0);                                                                        \
    RR_DMA(t + 2, 2)                                      /* slot B1 (free since X) */                    \
    RR_SBAR();                                                                                             \
    RR_PRIO(0)                                                                                             \
    RR_BLK(2, AF1, B0K1)                                                                                   \
    RR_SYNC(6)   /* Y: B1(t+1), A1(t+1) landed */       \
    RR_KEEP6()                                                                                             \
  }
  for (int t = 0; t < nk; ++t) RR_TILE()
#undef RR_TILE
  wait_vmcnt8<0>();

  // ---- next output tile: its first five half-tiles into ring slots 0-4 now; the epilogue stages through [80 KiB, 160 KiB)
  const int cm0 = m0, cn0 = n0;
  li += gstep;
  const bool has_next = li < chunk_n;
  if (has_next) {
    tile_origin(chunk0 + li, m0, n0);
    RR_SETUP_SRC(m0, n0)
    RR_DMA(0, 0) RR_DMA(0, 1) RR_DMA(0, 2) RR_DMA(0, 3)
    RR_DMA(1, 0)
  }

  // ---- epilogue of tile (cm0, cn0): v = acc * (scale * sa[m] * sw[n]) + bias[n] (+ GELU), 16-bit out, two 128-row passes
  {
  // per-lane addresses from an OPAQUE thread index, recomputed per tile: visible, the (tile-invariant) LDS addresses of the
  // parameter reads are hoisted out of the tile loop, kept live across the main loop at 256 VGPRs and spilled (gemm_kernel_hp)
  int tid_o_ = tid;
  asm volatile("" : "+v"(tid_o_));
  const int tid = tid_o_, lane = tid_o_ & 63, h = (tid_o_ & 63) >> 5;
  if constexpr (EPI == 3) {
    // e4m3 out: one byte per element, two 128-row passes through a [128][256 + 16] byte image.  A lane owns 4 consecutive
    // columns = one dword.  ds_write_b32 has 32 banks and its lane groups are 32 rows of one column: at the 272-byte pitch
    // rows r, r + 8, r + 16, r + 24 would meet on one bank (4-way), so the dword index inside every 16-byte chunk is XORed
    // with (r >> 3) & 3 and the reader undoes the permutation (first version: halves swapped on bit 4 only, 2-way,
    // SQ_LDS_BANK_CONFLICT 9.8 % of the kernel's LDS cycles).
    constexpr int PITCH = BN + 16, CPR = BN / 16, ROWS = 128;
    char* const stg = lds + 5 * HALF;
    uint8_t* const C8 = (uint8_t*)Cv;
    float rs[2][2];
#pragma unroll
    for (int hA = 0; hA < 2; ++hA)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        rs[hA][mb] = scale * *(const float*)(lds + PARAM_OFF + (hA * 128 + wr * 64 + mb * 32 + (lane & 31)) * 4);
      }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hA = q >> 1, hB = q & 1;
        if (hA != pass) continue;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cn = hB * 128 + wc * 32 + 8 * rg + 4 * h;
          const float4 bv = *(const float4*)(lds + PARAM_OFF + 2048 + cn * 4);      // (columns beyond N repeat the last one; never stored)
          const float4 cw = *(const float4*)(lds + PARAM_OFF + 1024 + cn * 4);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const float r_ = rs[hA][mb];
            const float v0 = fmaf(acc[q][mb][4 * rg + 0], r_ * cw.x, bv.x), v1 = fmaf(acc[q][mb][4 * rg + 1], r_ * cw.y, bv.y),
                        v2 = fmaf(acc[q][mb][4 * rg + 2], r_ * cw.z, bv.z), v3 = fmaf(acc[q][mb][4 * rg + 3], r_ * cw.w, bv.w);
            const f32x2 g0 = gelu_erf_fast2(f32x2{v0, v1}), g1 = gelu_erf_fast2(f32x2{v2, v3});
            int w8 = 0;                                       // (saturating: |gelu| * out_mul beyond 448 clamps)
            w8 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(g0.x * x.out_mul, -448.f, 448.f),
                                                 __builtin_amdgcn_fmed3f(g0.y * x.out_mul, -448.f, 448.f), w8, false);
            w8 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(g1.x * x.out_mul, -448.f, 448.f),
                                                 __builtin_amdgcn_fmed3f(g1.y * x.out_mul, -448.f, 448.f), w8, true);
            const int r = wr * 64 + mb * 32 + (lane & 31);
            *(uint32_t*)(stg + r * PITCH + (cn ^ (((r >> 3) & 3) << 2))) = (uint32_t)w8;
          }
        }
      }
      if (pass == 0 && has_next) wait_vmcnt8<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int row_base = cm0 + pass * 128;
      constexpr int UNR = ROWS * CPR / 512;
      static_assert(ROWS * CPR == 512 * UNR && UNR == 4, "one batch per pass");
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int i = tid + u * 512, r = i / CPR, c = i - r * CPR;
        const int gm = row_base + r, gcol = cn0 + c * 16;
        if (gm < M && gcol < N) {
          const uint4 v = *(const uint4*)(stg + r * PITCH + c * 16);
          // r = tid/16 + 32u: (r >> 3) & 3 = (tid >> 7) & 3 = k; dword j of the chunk holds columns 4 (j ^ k): two conditional swaps
          const bool k0 = (tid & 128) != 0, k1 = (tid & 256) != 0;
          const uint32_t a0 = k0 ? v.y : v.x, a1 = k0 ? v.x : v.y, a2 = k0 ? v.w : v.z, a3 = k0 ? v.z : v.w;      // j ^ 1
          store_stream(C8 + (size_t)gm * ldc + gcol, make_uint4(k1 ? a2 : a0, k1 ? a3 : a1, k1 ? a0 : a2, k1 ? a1 : a3));   // j ^ 2
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if constexpr (EPI == 4) {
    // fp32 out + residual: four 64-row passes (pass = 2 hA + wave row group) through a [64][1024 + 16] byte image.  A wave
    // streams one row per step (its LayerNorm statistics are a scalar load), a thread keeps ONE 4-column chunk for the whole
    // tile (gamma / beta loaded once); the 8 residual chunks of a pass are requested before the pass's staging writes.
    constexpr int PITCH = BN * 4 + 16, CPR = BN * 4 / 16, ROWS = 64, UNR = ROWS * CPR / 512;
    static_assert(ROWS * PITCH <= 5 * HALF && CPR == 64 && UNR == 8, "staging image above the five prefetch slots; one row per wave and step");
    char* const stg = lds + 5 * HALF;
    float* const C32 = (float*)Cv;
    const int c4 = tid & 63, gcol = cn0 + c4 * 4;
    const bool col_ok = gcol < N;
    float4 lg = make_float4(1.f, 1.f, 1.f, 1.f), lb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x.rstats) { lg = *(const float4*)(lds + PARAM_OFF + 3072 + c4 * 16); lb = *(const float4*)(lds + PARAM_OFF + 4096 + c4 * 16); }
    float rs[2][2];
#pragma unroll
    for (int hA = 0; hA < 2; ++hA)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        rs[hA][mb] = scale * *(const float*)(lds + PARAM_OFF + (hA * 128 + wr * 64 + mb * 32 + (lane & 31)) * 4);
      }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int hA = pass >> 1, wrp = pass & 1;
      const int row_base = cm0 + hA * 128 + wrp * 64;
      float4 xr[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int gm = row_base + wave + 8 * u;
        xr[u] = (gm < M && col_ok) ? load_stream_f4(x.resid + (size_t)gm * x.ldr + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (wr == wrp) {
#pragma unroll
        for (int hB = 0; hB < 2; ++hB) {
          const int q = 2 * hA + hB;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int cn = hB * 128 + wc * 32 + 8 * rg + 4 * h;
            RR_SBAR();      // (left free, hipcc hoists all sixteen parameter reads of the pass above the residual loads and spills)
            const float4 bv = *(const float4*)(lds + PARAM_OFF + 2048 + cn * 4);
            const float4 cw = *(const float4*)(lds + PARAM_OFF + 1024 + cn * 4);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
              const float r_ = rs[hA][mb];
              const int r = mb * 32 + (lane & 31);
              *(float4*)(stg + r * PITCH + cn * 4) =
                  make_float4(fmaf(acc[q][mb][4 * rg + 0], r_ * cw.x, bv.x), fmaf(acc[q][mb][4 * rg + 1], r_ * cw.y, bv.y),
                              fmaf(acc[q][mb][4 * rg + 2], r_ * cw.z, bv.z), fmaf(acc[q][mb][4 * rg + 3], r_ * cw.w, bv.w));
            }
            RR_SBAR();
          }
        }
      }
      if (pass == 0 && has_next) wait_vmcnt8<0>();           // (also the first pass's residual rows)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = wave + 8 * u, gm = row_base + r;
        if (gm < M && col_ok) {
          float4 xv = xr[u];
          if (x.rstats) {
            const float2 st2 = x.rstats[gm];
            xv = make_float4((xv.x - st2.x) * st2.y * lg.x + lb.x, (xv.y - st2.x) * st2.y * lg.y + lb.y,
                             (xv.z - st2.x) * st2.y * lg.z + lb.z, (xv.w - st2.x) * st2.y * lg.w + lb.w);
          }
          const float4 f = *(const float4*)(stg + r * PITCH + c4 * 16);
          store_stream(C32 + (size_t)gm * ldc + gcol, make_float4(f.x + xv.x, f.y + xv.y, f.z + xv.z, f.w + xv.w));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    constexpr int PITCH = BN * 2 + 16, CPR = BN * 2 / 16, ROWS = 128;
    static_assert(ROWS * PITCH <= 5 * HALF, "staging image must fit above the five prefetch slots");
    char* const stg = lds + 5 * HALF;
    float rs[2][2];                                          // row scales of the 4 rows this lane owns
#pragma unroll
    for (int hA = 0; hA < 2; ++hA)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        rs[hA][mb] = scale * *(const float*)(lds + PARAM_OFF + (hA * 128 + wr * 64 + mb * 32 + (lane & 31)) * 4);
      }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {                   // pass = hA: scale, bias (+ GELU), pack and stage this row half
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hA = q >> 1, hB = q & 1;
        if (hA != pass) continue;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cn = hB * 128 + wc * 32 + 8 * rg + 4 * h;
          const float4 bv = *(const float4*)(lds + PARAM_OFF + 2048 + cn * 4);      // (columns beyond N repeat the last one; never stored)
          const float4 cw = *(const float4*)(lds + PARAM_OFF + 1024 + cn * 4);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const float r_ = rs[hA][mb];
            float v0 = fmaf(acc[q][mb][4 * rg + 0], r_ * cw.x, bv.x), v1 = fmaf(acc[q][mb][4 * rg + 1], r_ * cw.y, bv.y),
                  v2 = fmaf(acc[q][mb][4 * rg + 2], r_ * cw.z, bv.z), v3 = fmaf(acc[q][mb][4 * rg + 3], r_ * cw.w, bv.w);
            if constexpr (EPI == 1 && RR_PK_GELU != 0) {          // packed Horner chain (rr_common.h): same values, half the VALU issue
              const f32x2 g0 = gelu_erf_fast2(f32x2{v0, v1}), g1 = gelu_erf_fast2(f32x2{v2, v3});
              v0 = g0.x; v1 = g0.y; v2 = g1.x; v3 = g1.y;
            } else if (EPI == 1) { v0 = gelu13(v0); v1 = gelu13(v1); v2 = gelu13(v2); v3 = gelu13(v3); }
            const int r = wr * 64 + mb * 32 + (lane & 31);
            // rows with bit 3 set keep the two 8-byte halves of every 16-byte chunk swapped (2-way write conflict at the
            // 528-byte pitch otherwise); the reader swaps them back — as in gemm_kernel_hp
            char* dst = stg + r * PITCH + ((cn * 2) ^ (lane & 8));
            *(uint2*)dst = make_uint2(pack2<DT>(v0, v1), pack2<DT>(v2, v3));
          }
        }
      }
      if (pass == 0 && has_next) wait_vmcnt8<0>();           // my pieces of the next tile's prefetch: before the first store
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int row_base = cm0 + pass * 128;
      constexpr int UNR = 8;
      static_assert(ROWS * CPR == 512 * UNR, "one batch per pass");
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int i = tid + u * 512, r = i / CPR, c = i - r * CPR;
        const int gm = row_base + r, gcol = cn0 + c * 8;
        if (gm < M && gcol < N) {
          uint4 v = *(const uint4*)(stg + r * PITCH + c * 16);
          if (tid & 256) v = make_uint4(v.z, v.w, v.x, v.y);   // r = tid/32 + 16u: bit 3 of r = bit 8 of tid
          store_stream((char*)C + ((size_t)gm * ldc + gcol) * 2, v);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // staging image consumed
    }
  }
  }   // opaque-index scope of the epilogue
  first_tile = false;
  if (!has_next) break;
  stage_params(m0, n0);                                     // the NEXT tile's (this tile's epilogue has read its own)
  RR_DMA(1, 1) RR_DMA(1, 2)                                 // slots 5, 6 were under the staging image until now
  }   // output tiles
#undef RR_SETUP_SRC
#undef RR_DMA
#undef RR_BLK
#undef RR_SBAR
#undef RR_PRIO
#undef RR_SYNC
#undef RR_KEEP6
}

// ---- per-tensor quantisation: y = e4m3(clamp(x / scale, +-448)), round to nearest even (v_cvt_pk_fp8_f32, OCP on gfx950)
template <bool F32_IN>
__global__ void quant_e4m3_kernel(const void* __restrict__ xv, float inv_scale, uint8_t* __restrict__ y, size_t n8) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float f[8];
  if constexpr (F32_IN) {
    const float4 a = ((const float4*)xv)[2 * i], b = ((const float4*)xv)[2 * i + 1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    const uint4 u = ((const uint4*)xv)[i];
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * inv_scale, -448.0f, 448.0f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  ((uint2*)y)[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
}

// max |x| of a tensor: non-negative floats order like their bit patterns, so an integer atomicMax is exact and
// independent of the order of arrival.  *out must be zeroed by the caller (rr_launch_amax does).
template <bool F32_IN>
__global__ void amax_kernel(const void* __restrict__ xv, size_t n8, unsigned int* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (F32_IN) {
      const float4 a = ((const float4*)xv)[2 * i], b = ((const float4*)xv)[2 * i + 1];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
    } else {
      const uint4 u = ((const uint4*)xv)[i];
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        m = fmaxf(m, fmaxf(fabsf(__uint_as_float(w[j] << 16)), fabsf(__uint_as_float(w[j] & 0xffff0000u))));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

}  // namespace

// x: n elements (n % 8 == 0) of f32 or of the 16-bit type bf16; y: n e4m3 bytes = e4m3(clamp(x / scale, +-448)).
hipError_t rr_launch_quant_e4m3(const void* x, int x_is_f32, float scale, uint8_t* y, size_t n, hipStream_t st) {
  if ((n & 7) || !(scale > 0.f)) return hipErrorInvalidValue;
  const size_t n8 = n >> 3;
  if (n8 == 0) return hipSuccess;
  const dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
  if (x_is_f32) hipLaunchKernelGGL((quant_e4m3_kernel<true>), grid, block, 0, st, x, 1.0f / scale, y, n8);
  else hipLaunchKernelGGL((quant_e4m3_kernel<false>), grid, block, 0, st, x, 1.0f / scale, y, n8);
  return hipGetLastError();
}

// *out (device float) = max |x|.
hipError_t rr_launch_amax(const void* x, int x_is_f32, size_t n, float* out, hipStream_t st) {
  if (n & 7) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  const size_t n8 = n >> 3;
  if (n8 == 0) return hipSuccess;
  const unsigned blocks = (unsigned)(n8 + 255) / 256 < 2048u ? (unsigned)((n8 + 255) / 256) : 2048u;
  if (x_is_f32) hipLaunchKernelGGL((amax_kernel<true>), dim3(blocks), dim3(256), 0, st, x, n8, (unsigned int*)out);
  else hipLaunchKernelGGL((amax_kernel<false>), dim3(blocks), dim3(256), 0, st, x, n8, (unsigned int*)out);
  return hipGetLastError();
}

// A8 [M,Kd] e4m3 bytes (row stride lda bytes), W8 [N,Kd] e4m3 (row stride ldw), bias [N] f32 or null; the accumulators are
// multiplied by scale * row_scale[m] * col_scale[n] (either vector may be null = 1); C: 16-bit in the operand type dt
// (epilogue 0, 1) or f32 (2), row stride ldc elements.  Kd % 128 == 0, N % 4 == 0.
// Epilogues 3 (erf-GELU -> e4m3 bytes scaled by out_mul, C = uint8 [M, ldc]) and 4 (f32 out + residual rows `resid` [M, ldr],
// LayerNorm-recomputed from rstats / rgamma / rbeta when rstats != null) exist on the persistent ring only: rr_gemm_fp8_ring_ok.
bool rr_gemm_fp8_ring_ok(int M, int N, int Kd) {
  return M > 0 && N > 0 && Kd > 0 && !(Kd % BKB) && !(N & 15) && ((M + 255) / 256) * ((N + 255) / 256) >= 512;
}
hipError_t rr_launch_gemm_fp8(const uint8_t* A, int lda, const uint8_t* W, int ldw, const float* bias, float scale,
                              const float* row_scale, const float* col_scale, void* C, int ldc, int M, int N, int Kd,
                              int epilogue, int dt, hipStream_t st, float out_mul, const float* resid, int ldr,
                              const float* rstats, const float* rgamma, const float* rbeta) {
  if (M <= 0 || N <= 0 || Kd <= 0 || (Kd % BKB) || (N & 3) || (lda & 15) || (ldw & 15) || (ldc & 3)) return hipErrorInvalidValue;
  if (epilogue < 0 || epilogue > 4 || (dt != 0 && dt != 1)) return hipErrorInvalidValue;
  if (epilogue >= 3) {
    if (!rr_gemm_fp8_ring_ok(M, N, Kd)) return hipErrorInvalidValue;
    if (epilogue == 3 && (!(out_mul > 0.f) || (ldc & 15))) return hipErrorInvalidValue;
    if (epilogue == 4 && (!resid || (ldr & 3) || (rstats && (!rgamma || !rbeta)))) return hipErrorInvalidValue;
  }
  F8Extra ex;
  ex.out_mul = out_mul;
  ex.resid = resid;
  ex.ldr = ldr;
  ex.rstats = (const float2*)rstats;
  ex.rgamma = rgamma;
  ex.rbeta = rbeta;
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (nwg >= 512 && !(N & 7) && epilogue != 2) {           // production: persistent ring, one workgroup per CU
    static std::atomic<int> cus[64];
    int n_cu = (dev >= 0 && dev < 64) ? cus[dev].load() : 0;
    if (!n_cu) {
      hipDeviceProp_t prop;
      if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
      n_cu = prop.multiProcessorCount & ~7;
      if (dev >= 0 && dev < 64) cus[dev].store(n_cu);
    }
    if (n_cu < 8) return hipErrorInvalidValue;
    constexpr int lds_bytes = 160 * 1024;
    const dim3 grid((unsigned)n_cu), block(512);
#define RR_HP8(E, D)                                                                                                  \
  {                                                                                                                  \
    auto kern = gemm_kernel_hp8<E, D>;                                                                               \
    static std::atomic<unsigned long long> mask{0};                                                                  \
    if (!(dev >= 0 && dev < 64 && ((mask.load() >> dev) & 1ull))) {                                                  \
      if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e; \
      if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev);                                                          \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, A, lda, W, ldw, bias, scale, row_scale, col_scale, C, ldc, M, N,   \
                       Kd, tiles_n, nwg, ex);                                                                        \
  }
    if (epilogue == 0) { if (dt == 0) RR_HP8(0, 0) else RR_HP8(0, 1) }
    else if (epilogue == 1) { if (dt == 0) RR_HP8(1, 0) else RR_HP8(1, 1) }
    else if (epilogue == 3) RR_HP8(3, 0)                    /* (the operand type does not enter an e4m3 / f32 output) */
    else RR_HP8(4, 0)
#undef RR_HP8
    return hipGetLastError();
  }
  constexpr int lds_bytes = 2 * (BM + BN) * BKB;
  const dim3 grid_exact((unsigned)nwg), block(WM * WN * 64);     // the XCD tile map is a bijection on [0, nwg)
#define RR_F8(E, D)                                                                                                  \
  {                                                                                                                  \
    auto kern = gemm_kernel_f8<BM, BN, WM, WN, E, D>;                                                                \
    static std::atomic<unsigned long long> mask{0};                                                                  \
    if (!(dev >= 0 && dev < 64 && ((mask.load() >> dev) & 1ull))) {                                                  \
      if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e; \
      if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev);                                                          \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid_exact, block, lds_bytes, st, A, lda, W, ldw, bias, scale, row_scale, col_scale, C, ldc, M, N, \
                       Kd, tiles_n, nwg);                                                                            \
  }
  switch (epilogue * 2 + dt) {
    case 0: RR_F8(0, 0) break;
    case 1: RR_F8(0, 1) break;
    case 2: RR_F8(1, 0) break;
    case 3: RR_F8(1, 1) break;
    case 4: RR_F8(2, 0) break;
    default: RR_F8(2, 1) break;
  }
#undef RR_F8
  return hipGetLastError();
}
