// e4m3 (OCP fp8) GEMM for gfx950 on the block-scaled matrix core:  C[M,N] = epi(scale * A8[M,K] · W8[N,K]^T + bias)
//
// First building block of BASELINE configs[4] ("bert-large ... fp8 MFMA"): per-tensor-scaled e4m3
// operands for the big GEMMs.  On gfx950 the 2x rate only exists on v_mfma_scale_f32_16x16x128_f8f6f4 (the plain
// _fp8_fp8 forms run at the bf16 rate, MI355X_MICROARCH.md "FP8"), so this kernel uses it with all block scales fixed at
// 2^0 (E8M0 127) and applies the per-tensor scale sa*sw in the epilogue.  Operand lane map (tools/fp8_mfma_probe.hip,
// exact integer data on the device): lane l holds A[row l&15][k = 32 (l>>4) + j], j = 0..31 in byte order, likewise B;
// C/D as every 16x16 MFMA (col = l&15, row = 4 (l>>4) + reg).
//
// Structure = the simple 16-bit kernel (gemm_bf16.hip, variant S): a 128-byte K-tile row holds 128 fp8 instead of 64
// bf16, so the LDS images, the LDS-DMA pieces and the swizzle are byte-for-byte the same; a lane's fragment is the two
// 16-byte chunks 2g, 2g+1 of its row, and one K-tile is ONE MFMA per 16x16 output block (twice the cycles of the bf16
// form, four times the K).  Not yet the persistent ring (gemm_kernel_hp) — that port is the next step.
#include "rr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int BKB = 128;     // K-tile: 128 bytes = 128 fp8 per row

template <int N>
__device__ __forceinline__ void wait_vmcnt8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float gelu13(float x) {   // erf-GELU, the 13-operation form of gemm_bf16.hip:gelu_fast
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f((ax * ax) * (-0.5f * 1.4426950408889634f));
  return fmaf(-ax, (p * t) * e, __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_huge_valf()));
}

// EPI: 0 = bf16(acc*s + b), 1 = bf16(gelu(acc*s + b)), 2 = f32(acc*s + b)
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel_f8(const uint8_t* __restrict__ A, int lda,
                                                              const uint8_t* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, float scale,
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
    a_off[i] = (uint32_t)((size_t)min(m0 + r, M - 1) * lda + c * 16);
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = (wave * PW + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w_off[i] = (uint32_t)((size_t)min(n0 + r, N - 1) * ldw + c * 16);
  }
  const uint32_t lds_base = lds_addr(lds);
  auto stage = [&](int buf, int k0) {
    const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + wave * PA * 1024);
    const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + A_BYTES + wave * PW * 1024);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16_so(A + k0, a_off[i], a_dst + i * 1024);
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16_so(W + k0, w_off[i], w_dst + i * 1024);
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
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gm = m0 + wm * TM + mt * 16 + (lane & 15);
      if (gm >= M) continue;
      float v0 = fmaf(acc[nt][mt][0], scale, bv.x), v1 = fmaf(acc[nt][mt][1], scale, bv.y),
            v2 = fmaf(acc[nt][mt][2], scale, bv.z), v3 = fmaf(acc[nt][mt][3], scale, bv.w);
      if (EPI == 1) { v0 = gelu13(v0); v1 = gelu13(v1); v2 = gelu13(v2); v3 = gelu13(v3); }
      if (EPI == 2) *(float4*)((float*)Cv + (size_t)gm * ldc + gn) = make_float4(v0, v1, v2, v3);
      else *(uint2*)((bf16_t*)Cv + (size_t)gm * ldc + gn) = make_uint2(pack2<0>(v0, v1), pack2<0>(v2, v3));
    }
  }
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

// A8 [M,Kd] e4m3 bytes (row stride lda bytes), W8 [N,Kd] e4m3 (row stride ldw), bias [N] f32 or null, scale = sa * sw
// (per-tensor dequantisation), C: bf16 (epilogue 0, 1) or f32 (2), row stride ldc elements.  Kd % 128 == 0, N % 4 == 0.
hipError_t rr_launch_gemm_fp8(const uint8_t* A, int lda, const uint8_t* W, int ldw, const float* bias, float scale, void* C,
                              int ldc, int M, int N, int Kd, int epilogue, hipStream_t st) {
  if (M <= 0 || N <= 0 || Kd <= 0 || (Kd % BKB) || (N & 3) || (lda & 15) || (ldw & 15) || (ldc & 3)) return hipErrorInvalidValue;
  if (epilogue < 0 || epilogue > 2) return hipErrorInvalidValue;
  if ((size_t)M * lda >= (1ull << 32) || (size_t)N * ldw >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit DMA offsets
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
  constexpr int lds_bytes = 2 * (BM + BN) * BKB;
  const dim3 grid_exact((unsigned)nwg), block(WM * WN * 64);     // the XCD tile map is a bijection on [0, nwg)
#define RR_F8(E)                                                                                                     \
  {                                                                                                                  \
    auto kern = gemm_kernel_f8<BM, BN, WM, WN, E>;                                                                   \
    static bool attr_done = false;                                                                                   \
    if (!attr_done) {                                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);  \
      if (e != hipSuccess) return e;                                                                                 \
      attr_done = true;                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid_exact, block, lds_bytes, st, A, lda, W, ldw, bias, scale, C, ldc, M, N, Kd, tiles_n, nwg); \
  }
  switch (epilogue) {
    case 0: RR_F8(0) break;
    case 1: RR_F8(1) break;
    default: RR_F8(2) break;
  }
#undef RR_F8
  return hipGetLastError();
}
