// Probe: what the chip SUSTAINS, at the clock it holds under its power cap, for the two fp16 matrix instructions of equal peak rate —
// v_mfma_f32_16x16x32_f16 (the GEMM kernels' instruction: 2 x 16-byte operands per 16 384 FLOP) and v_mfma_f32_32x32x16_f16 (2 x 16-byte
// operands per 32 768 FLOP: half the register-operand bytes per FLOP) — on RANDOM operands, every CU busy, two waves per SIMD, with and
// without a stream of LDS fragment reads beside the MFMAs (12 ds_read_b128 per 32 MFMA-equivalents, the GEMM main loop's ratio).
// Question (docs/rounds/r05.md): every hot kernel here is power-limited (1.7 GHz under load, 2.3 GHz on half the CUs); does the
// instruction shape change the sustained rate?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe && tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 tile[4096];      // 64 KiB of operand bytes
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 512) tile[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  // 12 operand fragments per lane, random fp16 values in [-1, 1)
  f16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(f16x8, tile[(tid * 13 + i * 517) & 4095]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(f16x8, tile[(tid * 7 + i * 911 + 3) & 4095]);
  float keep = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if constexpr (LDS) {      // refresh the 12 fragments from LDS, as one 32-deep k-step of the GEMM main loop does
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(f16x8, tile[(tid + it * 64 + i * 512) & 4095]);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(f16x8, tile[(tid + it * 64 + i * 512 + 256) & 4095]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if constexpr (LDS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(f16x8, tile[(tid + it * 64 + i * 512) & 4095]);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(f16x8, tile[(tid + it * 64 + i * 512 + 256) & 4095]);
      }
      // the same 128 x 64 x 32 wave tile: 4 x 2 tiles of 32 x 32, two 16-deep k-steps (fragments a[2i], a[2i+1] / b[2j], b[2j+1])
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[2 * j + k], a[2 * i + k], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) keep += acc[i][j][0] + acc[i][j][15];
  }
  if (keep == 1.2345e-30f) sink[blockIdx.x * 512 + tid] = keep;
}

// One wave per SIMD (256-thread workgroup, up to 512 registers per lane): a 128 x 128 x 32 wave tile — 64 MFMAs of 16x16x32 on 256
// accumulator registers, 16 fragment reads per iteration = 8 ds_read_b128 per 32 MFMAs against the two-waves-per-SIMD tile's 12.
template <bool LDS>
__global__ __launch_bounds__(256) void probe_w4(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 tile[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) tile[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  f16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(f16x8, tile[(tid * 13 + i * 517) & 4095]);
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(f16x8, tile[(tid * 7 + i * 911 + 3) & 4095]);
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if constexpr (LDS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(f16x8, tile[(tid + it * 64 + i * 512) & 4095]);
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(f16x8, tile[(tid + it * 64 + i * 512 + 256) & 4095]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  float keep = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][3];
  if (keep == 1.2345e-30f) sink[blockIdx.x * 256 + tid] = keep;
}

int main() {
  const int n = 1 << 20;
  std::vector<uint16_t> h(n * 8);
  uint32_t s = 12345;
  for (auto& v : h) {        // fp16 in [-1, 1): random sign / mantissa, exponent 2^-1 .. 2^-4
    s = s * 1664525u + 1013904223u;
    v = (uint16_t)(((s >> 16) & 0x83ff) | ((11 + ((s >> 27) & 3)) << 10));
  }
  uint4* d; float* sink;
  hipMalloc(&d, n * 16); hipMalloc(&sink, 256 * 512 * 4);
  hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256;
  auto run = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, d, sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = 2.0 * 128 * 64 * 32 * (double)iters * 8 * grid;      // one 128 x 64 x 32 wave tile per iteration and wave
      const double cyc_per_it_at_peak = 32.0 * 16.0 * 2;                     // 32 16x16x32 MFMAs of 16 cycles, two waves per SIMD
      printf("%-34s rep %d: %8.2f ms  %7.1f TFLOP/s  (clock if the matrix pipe were always busy: %.2f GHz)\n", name, rep, ms, fl / ms / 1e9,
             iters * cyc_per_it_at_peak / (ms * 1e-3) / 1e9);
    }
  };
  auto run4 = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = 2.0 * 128 * 128 * 32 * (double)iters * 4 * grid;     // one 128 x 128 x 32 wave tile per iteration and wave, four waves per CU
      printf("%-34s rep %d: %8.2f ms  %7.1f TFLOP/s  (clock if the matrix pipe were always busy: %.2f GHz)\n", name, rep, ms, fl / ms / 1e9,
             iters * 64.0 * 16.0 / (ms * 1e-3) / 1e9);
    }
  };
  run(probe<16, false>, "16x16x32 f16, registers only");
  run(probe<32, false>, "32x32x16 f16, registers only");
  run(probe<16, true>, "16x16x32 f16 + 12 ds_read_b128");
  run(probe<32, true>, "32x32x16 f16 + 12 ds_read_b128");
  run(probe<16, false>, "16x16x32 f16, registers only");
  run(probe<32, false>, "32x32x16 f16, registers only");
  run4(probe_w4<false>, "1 wave/SIMD 128x128, registers only");
  run4(probe_w4<true>, "1 wave/SIMD 128x128 + 16 ds_read");
  run(probe<16, true>, "16x16x32 f16 + 12 ds_read_b128");
  run4(probe_w4<true>, "1 wave/SIMD 128x128 + 16 ds_read");
  return 0;
}
