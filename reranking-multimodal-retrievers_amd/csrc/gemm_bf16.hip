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
//   * 128x128x64 tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA tiles);
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): the LDS image is lane-linear,
//     the XOR bank swizzle is applied on the per-lane SOURCE address and again on the ds_read_b128
//     (cdna_hip_programming.md §5.4 rule 21); double-buffered, one barrier per 64-deep K step, the
//     next stage's DMA is in flight under the current stage's 32 MFMAs per wave;
//   * the MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends up with 4 consecutive output columns of one output row: bias/GELU/residual epilogues
//     work on float4 and stores are 8/16-byte vectors;
//   * blockIdx -> tile map is XCD-aware (bijective remap, T1): the workgroups that share an
//     activation row-panel run on one XCD so the panel is fetched from HBM once per XCD L2.
#include "rr_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;             // 16 KiB
constexpr int W_BYTES = BN * BK * 2;             // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;   // 32 KiB, x2 buffers = 64 KiB / workgroup

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm128_kernel(const bf16_t* __restrict__ A, int lda,
                                                      const bf16_t* __restrict__ W, int ldw,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ resid, int ldr,
                                                      void* __restrict__ Cv, int ldc, int M, int N, int Kd,
                                                      int tiles_n, int nwg) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];

  // ---- XCD-aware bijective block remap: blocks b, b+8, ... share an XCD (round-robin dispatch);
  // give each XCD a contiguous run of tiles.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;   // n fastest: A panel reused from L2
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging: each wave issues 4 A pieces + 4 W pieces (1 KiB = 8 rows x 128 B each) per stage.
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);              // source chunk that lands in slot lane&7
    const int gm = min(m0 + r, M - 1), gn = min(n0 + r, N - 1);   // clamp: padded rows re-read a valid row
    a_src[i] = A + (size_t)gm * lda + c * 8;
    w_src[i] = W + (size_t)gn * ldw + c * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* a_dst = lds + buf * STAGE_BYTES + wave * 4 * 1024;
    char* w_dst = a_dst + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + k0, a_dst + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + k0, w_dst + i * 1024);
  };

  f32x4 acc[4][4];   // [nt][mt]; lane holds m = mt*16 + (lane&15), n = nt*16 + (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = Kd / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of stage kt landed
    __syncthreads();                                     // ... everyone's; and compute(kt-1) is done
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* a_t = lds + (kt & 1) * STAGE_BYTES;
    const char* w_t = a_t + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
      const int c = ks * 4 + (lane >> 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *(const bf16x8*)(a_t + swz128(wm * 64 + t * 16 + (lane & 15), c));
        wf[t] = *(const bf16x8*)(w_t + swz128(wn * 64 + t * 16 + (lane & 15), c));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    }
  }

  // ---- epilogue
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int gn = n0 + wn * 64 + nt * 16 + (lane >> 4) * 4;
    if (gn >= N) continue;
    float4 bv = bias ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int gm = m0 + wm * 64 + mt * 16 + (lane & 15);
      if (gm >= M) continue;
      float v0 = acc[nt][mt][0] + bv.x, v1 = acc[nt][mt][1] + bv.y, v2 = acc[nt][mt][2] + bv.z,
            v3 = acc[nt][mt][3] + bv.w;
      if (EPI == EPI_BIAS_GELU_BF16) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
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

}  // namespace

hipError_t rr_launch_gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias,
                          const float* resid, int ldr, void* C, int ldc, int M, int N, int Kd,
                          int epilogue, hipStream_t st) {
  if (M <= 0 || N <= 0 || Kd <= 0) return hipErrorInvalidValue;
  if (Kd % BK != 0 || (lda & 7) || (ldw & 7) || (N & 3) || (ldc & 3)) return hipErrorInvalidValue;
  if (epilogue == EPI_BIAS_RESID_F32 && (!resid || (ldr & 3))) return hipErrorInvalidValue;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
  dim3 grid(nwg), block(256);
#define RR_GEMM_CASE(E)                                                                            \
  case E:                                                                                          \
    hipLaunchKernelGGL(gemm128_kernel<E>, grid, block, 0, st, A, lda, W, ldw, bias, resid, ldr, C, \
                       ldc, M, N, Kd, tiles_n, nwg);                                               \
    break;
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
