// Probe: how many bytes per clock does ONE CU's vector-memory pipeline move, by instruction kind and access shape, with every
// CU busy (one 512-thread workgroup per CU, 8 waves, each wave keeps DEPTH wave-instructions of 1 KiB in flight)?
// Round 4's direct-epilogue GEMM (removed) suggested one constant for LDS-DMA, register loads and stores alike (~20-25 B/clk/CU)
// and a 2x penalty for stores of 16 rows x 64 B against whole 512-byte row segments; this measures it in isolation.
//   mode 0: LDS-DMA (global_load_lds_dwordx4), 8 rows x 128 B per instruction, row pitch 1536 B (the GEMM's operand piece)
//   mode 1: LDS-DMA, 1 KiB contiguous per instruction
//   mode 2: global_load_dwordx4 to registers, 8 rows x 128 B, pitch 1536 B
//   mode 3: global_load_dwordx4 to registers, 1 KiB contiguous
//   mode 4: global_store_dwordx4, 2 rows x 512 B (the staged epilogue's stream-out), pitch 1536 B
//   mode 5: global_store_dwordx4, 16 rows x 64 B (the direct epilogue's accumulator layout), pitch 1536 B
//   mode 6: as 4 with the nt policy;  mode 7: as 5 with the nt policy
// Footprint per workgroup: `span` bytes walked cyclically (64 KiB: stays in the XCD's L2; 8 MiB: streams from beyond it).
//   hipcc --offload-arch=gfx950 -O3 tools/mem_pipeline_probe.hip -o tools/bin/mem_pipeline_probe && tools/bin/mem_pipeline_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(char* buf, size_t span, int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* base = buf + (size_t)blockIdx.x * span;
  // per-lane offset inside one 1 KiB-payload instruction
  uint32_t lane_off;
  constexpr int PITCH = 1536;
  if (MODE == 0 || MODE == 2) lane_off = (lane >> 3) * PITCH + (lane & 7) * 16;          // 8 rows x 128 B
  else if (MODE == 1 || MODE == 3) lane_off = lane * 16;                                 // 1 KiB contiguous
  else if (MODE == 4 || MODE == 6) lane_off = (lane >> 5) * PITCH + (lane & 31) * 16;    // 2 rows x 512 B
  else lane_off = (lane & 15) * PITCH + (lane >> 4) * 16;                                // 16 rows x 64 B
  // an instruction's footprint in the buffer (rows x pitch); instructions of a wave and of the 8 waves tile the span
  const uint32_t foot = (MODE == 1 || MODE == 3) ? 1024u : (MODE == 0 || MODE == 2) ? 8u * PITCH : (MODE == 4 || MODE == 6) ? 2u * PITCH : 16u * PITCH;
  // the 8 waves side by side: below each other, except the 16-row form whose waves take neighbouring 64-byte columns of the
  // same 16 rows (as the four column groups x two column blocks of a GEMM wave row do)
  const bool cols = MODE == 5 || MODE == 7;
  const uint32_t per_round = cols ? foot : foot * 8, wave_off = cols ? wave * 64u : wave * foot;
  const uint32_t rounds = (uint32_t)(span / per_round);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * (DEPTH * 1024);
  u32x4 r[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) r[d] = u32x4{(uint32_t)lane, 1u, 2u, (uint32_t)d};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  uint32_t rd = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint32_t off = (rd % rounds) * per_round + wave_off + lane_off;
      ++rd;
      if (MODE <= 1) {
        glds16(base, off, __builtin_amdgcn_readfirstlane(lds_base + d * 1024));
      } else if (MODE <= 3) {
        // "+v": the destination stays THIS register for the whole loop — as a plain output hipcc may hand the register to
        // something else (an offset, say) while the load is still in flight and the late data would corrupt it
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(r[d]) : "v"(off), "s"(base) : "memory");
      } else if (MODE <= 5) {
        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(r[d]), "s"(base) : "memory");
      } else {
        asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(r[d]), "s"(base) : "memory");
      }
    }
    // keep DEPTH instructions in flight: wait until only the youngest DEPTH/2 are outstanding
    if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float acc = 0.f;
  if (MODE == 2 || MODE == 3) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += (float)r[d][0];
  } else if (MODE <= 1) {
    acc = *(float*)(lds + threadIdx.x * 4);
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE, int DEPTH>
void run(const char* name, char* buf, size_t span, unsigned long long* cyc, float* sink) {
  const int grid = 256, iters = 400;
  auto kern = probe<MODE, DEPTH>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  std::vector<unsigned long long> h(grid);
  double med_cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 8 * DEPTH * 1024, 0, buf, span, iters, cyc, sink);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) {
      best = ms;
      hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      med_cyc = (double)h[grid / 2];
    }
  }
  const double bytes_per_cu = (double)iters * DEPTH * 8 * 1024;
  printf("%-58s depth %2d span %5zu KiB: %6.1f B/clk/CU (median workgroup), %6.1f GB/s/CU, %5.2f TB/s chip, clock %.2f GHz\n", name, DEPTH,
         span >> 10, bytes_per_cu / med_cyc, bytes_per_cu / (best * 1e-3) / 1e9, bytes_per_cu * grid / (best * 1e-3) / 1e12,
         med_cyc / (best * 1e-3) / 1e9);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t big = 8u << 20;
  char* buf; unsigned long long* cyc; float* sink;
  hipMalloc(&buf, 256 * big + (1 << 20)); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
  hipMemset(buf, 1, 256 * big);
  for (size_t span : {(size_t)(96u << 10), big}) {
    printf("---- footprint per workgroup %zu KiB (%s)\n", span >> 10, span < (1u << 20) ? "L2-resident" : "streams from beyond L2");
    run<0, 8>("LDS-DMA, 8 rows x 128 B (GEMM operand piece)", buf, span, cyc, sink);
    run<0, 16>("LDS-DMA, 8 rows x 128 B", buf, span, cyc, sink);
    run<1, 8>("LDS-DMA, 1 KiB contiguous", buf, span, cyc, sink);
    run<2, 8>("load to registers, 8 rows x 128 B", buf, span, cyc, sink);
    run<3, 8>("load to registers, 1 KiB contiguous", buf, span, cyc, sink);
    run<4, 8>("store, 2 rows x 512 B (staged epilogue)", buf, span, cyc, sink);
    run<5, 8>("store, 16 rows x 64 B (accumulator layout)", buf, span, cyc, sink);
    run<6, 8>("store nt, 2 rows x 512 B", buf, span, cyc, sink);
    run<7, 8>("store nt, 16 rows x 64 B", buf, span, cyc, sink);
  }
  return 0;
}
