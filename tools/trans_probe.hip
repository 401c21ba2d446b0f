// Issue-rate probe: how many cycles does a wave spend per v_exp_f32, per v_fma_f32, and per pair when they alternate?
// (is the transcendental unit a separate pipe that runs beside plain VALU work of the SAME wave / of ANOTHER wave?)
//   hipcc --offload-arch=gfx950 -O3 tools/trans_probe.hip -o tools/bin/trans_probe && tools/bin/trans_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: exp only, 1: fma only, 2: exp, fma alternating (1:1), 3: exp + 2 fma, 4: exp + 4 fma, 5: rcp only
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 0.5f + i * 0.1f; }
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0 || MODE >= 2 && MODE <= 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 1 || MODE == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[i]));
      if (MODE == 3) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[(i + 4) & 7])); }
      if (MODE == 4) {
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[(i + 2) & 7]));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[(i + 4) & 7])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[(i + 6) & 7]));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int per_iter_exp, int per_iter_fma) {
  const int blocks = 256 * waves_per_simd, iters = 2000;      // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  probe<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  probe<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += v; m /= blocks;
  printf("%-26s %d wave(s)/SIMD: %7.2f cycles per inner step (%d exp/rcp + %d fma per wave)\n", name, waves_per_simd,
         m / (iters * 8.0), per_iter_exp, per_iter_fma);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_exp_f32", w, 1, 0);
    run<5>("v_rcp_f32", w, 1, 0);
    run<1>("v_fma_f32", w, 0, 1);
    run<2>("exp + 1 fma", w, 1, 1);
    run<3>("exp + 2 fma", w, 1, 2);
    run<4>("exp + 4 fma", w, 1, 4);
  }
  return 0;
}
