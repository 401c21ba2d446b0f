// Probe: what does s_barrier cost an MFMA-bound loop when ONE workgroup owns the CU?
// Loop body = NM independent-accumulator v_mfma_f32_16x16x32_bf16 per wave, then (mode) nothing / s_barrier /
// s_barrier every 2nd iteration / LDS arrive + poll.  Prints cycles per iteration (median over workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NM, int MODE>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  volatile uint32_t* cnt = (volatile uint32_t*)lds;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
  f32x4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nw = blockDim.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NM / 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1) __builtin_amdgcn_s_barrier();
    if (MODE == 2 && (it & 1)) __builtin_amdgcn_s_barrier();
    if (MODE == 3) {   // arrive + immediate poll on an LDS counter
      if (lane == 0) __hip_atomic_fetch_add((uint32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t want = (uint32_t)nw * (uint32_t)(it + 1);
      for (int s = 0; s < (1 << 14); ++s)
        if (__builtin_amdgcn_readfirstlane(*cnt) >= want) break;
    }
    if (MODE == 4) {   // arrive now, poll for the PREVIOUS iteration's arrivals (one iteration of slack)
      if (lane == 0) __hip_atomic_fetch_add((uint32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t want = (uint32_t)nw * (uint32_t)it;
      for (int s = 0; s < (1 << 14); ++s)
        if (__builtin_amdgcn_readfirstlane(*cnt) >= want) break;
    }
    if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 6) asm volatile("s_sleep 2" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 256 * 8); hipMalloc(&sink, 4);
  const int iters = 2000;
  auto run = [&](auto kern, const char* name, int threads) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    std::vector<unsigned long long> h(256);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 131072, 0, iters, out, sink);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-46s threads %3d: %7.1f cycles/iter (median), min %7.1f max %7.1f\n", name, threads, (double)h[128] / iters,
           (double)h[0] / iters, (double)h[255] / iters);
  };
  for (int threads : {512, 256}) {
    run(probe<16, 0>, "16 MFMA, no sync", threads);
    run(probe<16, 1>, "16 MFMA, s_barrier", threads);
    run(probe<16, 2>, "16 MFMA, s_barrier every 2nd", threads);
    run(probe<16, 3>, "16 MFMA, LDS arrive+poll", threads);
    run(probe<16, 4>, "16 MFMA, LDS arrive, poll previous", threads);
    run(probe<16, 5>, "16 MFMA, lgkmcnt(0)", threads);
    run(probe<16, 6>, "16 MFMA, s_sleep 2", threads);
    run(probe<32, 0>, "32 MFMA, no sync", threads);
    run(probe<32, 1>, "32 MFMA, s_barrier", threads);
    run(probe<64, 0>, "64 MFMA, no sync", threads);
    run(probe<64, 1>, "64 MFMA, s_barrier", threads);
    run(probe<8, 0>, "8 MFMA, no sync", threads);
    run(probe<8, 1>, "8 MFMA, s_barrier", threads);
  }
  return 0;
}
