// Probe: operand lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950 (exact small-integer data).
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_mfma32_probe.hip -o /tmp/fp8_probe32 && /tmp/fp8_probe32
// Hypothesis (by analogy with mfma_f32_32x32x16_bf16: A[row l&31][k = 8 (l>>5) + j]): lane l holds
//   A[row l&31][k = 32 (l>>5) + j] and B[k = 32 (l>>5) + j][col l&31], j = 0..31 in byte order;
//   C/D as every 32x32 MFMA: col = l&31, row = (reg&3) + 8 (reg>>2) + 4 (l>>5).  Block scales: E8M0, 127 = 2^0.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static uint8_t enc_e4m3(float v) {
  if (v == 0.f) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  float a = std::fabs(v);
  int e = (int)std::floor(std::log2(a));
  float m = a / std::ldexp(1.0f, e) - 1.0f;
  int mi = (int)std::lround(m * 8);
  return s | (uint8_t)(((e + 7) << 3) | mi);
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int scale) {   // A [32][64], B stored [n][k] = [32][64]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8 a, b;
  const int* ap = (const int*)(A + r * 64 + 32 * h);
  const int* bp = (const int*)(B + r * 64 + 32 * h);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];     // row, col = r
}

int main() {
  const float vals[] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, -1.f, -2.f, -0.5f, 6.f, -3.f};
  std::vector<uint8_t> A(32 * 64), B(32 * 64);
  std::vector<float> Af(32 * 64), Bf(32 * 64), ref(1024, 0.f), out(1024);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 16) % 12; };
  for (int i = 0; i < 32 * 64; ++i) { Af[i] = vals[rnd()]; A[i] = enc_e4m3(Af[i]); Bf[i] = vals[rnd()]; B[i] = enc_e4m3(Bf[i]); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float acc = 0; for (int k = 0; k < 64; ++k) acc += Af[m * 64 + k] * Bf[n * 64 + k]; ref[m * 32 + n] = acc; }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int scale : {127, 0x7f7f7f7f}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, scale);
    hipMemcpy(out.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
    double maxd = 0; int bad = 0, badT = 0;
    for (int i = 0; i < 1024; ++i) { double d = std::fabs(out[i] - ref[i]); if (d > 1e-3) ++bad; if (d > maxd) maxd = d; }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) if (std::fabs(out[n * 32 + m] - ref[m * 32 + n]) > 1e-3) ++badT;
    printf("scale=0x%x: mismatches %d / 1024 (as D^T: %d), max |d| %.3f, D[0][0..3] = %.2f %.2f %.2f %.2f  ref %.2f %.2f %.2f %.2f\n",
           scale, bad, badT, maxd, out[0], out[1], out[2], out[3], ref[0], ref[1], ref[2], ref[3]);
  }
  return 0;
}
