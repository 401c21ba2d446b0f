// Probe: operand lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands on gfx950 (exact small-integer data).
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_mfma_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
// Hypothesis: lane l holds A[row l&15][k = 32 (l>>4) + j] and B[k = 32 (l>>4) + j][col l&15], j = 0..31 in byte order;
// C/D as every 16x16 MFMA: col = l&15, row = 4 (l>>4) + reg.  Block scales: E8M0, 127 = 2^0.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static uint8_t enc_e4m3(float v) {   // exact for the values used here (|v| in {0, 0.5, 1, 1.5, 2, 3, 4, 6, 8})
  if (v == 0.f) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  float a = std::fabs(v);
  int e = (int)std::floor(std::log2(a));
  float m = a / std::ldexp(1.0f, e) - 1.0f;          // [0, 1)
  int mi = (int)std::lround(m * 8);
  return s | (uint8_t)(((e + 7) << 3) | mi);
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int scale) {   // A [16][128], B stored [n][k] = [16][128]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 a, b;
  const int* ap = (const int*)(A + r * 128 + 32 * g);
  const int* bp = (const int*)(B + r * 128 + 32 * g);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];     // row = 4g + i, col = r
}

int main() {
  const float vals[] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, -1.f, -2.f, -0.5f, 6.f, -3.f};
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  std::vector<float> Af(16 * 128), Bf(16 * 128), ref(256, 0.f), out(256);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 16) % 12; };
  for (int i = 0; i < 16 * 128; ++i) { Af[i] = vals[rnd()]; A[i] = enc_e4m3(Af[i]); Bf[i] = vals[rnd()]; B[i] = enc_e4m3(Bf[i]); }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float acc = 0; for (int k = 0; k < 128; ++k) acc += Af[m * 128 + k] * Bf[n * 128 + k]; ref[m * 16 + n] = acc; }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int scale : {127, 0x7f7f7f7f, 0}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, scale);
    hipMemcpy(out.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    double maxd = 0, ratio = 0; int bad = 0;
    for (int i = 0; i < 256; ++i) { double d = std::fabs(out[i] - ref[i]); if (d > 1e-3) ++bad; if (d > maxd) maxd = d; if (ref[i] != 0) ratio = out[i] / ref[i]; }
    // also the transposed reading of D
    int badT = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) if (std::fabs(out[n * 16 + m] - ref[m * 16 + n]) > 1e-3) ++badT;
    printf("scale=0x%x: mismatches %d / 256 (as D^T: %d), max |d| %.3f, last ratio %.4g, D[0][0..3] = %.2f %.2f %.2f %.2f  ref %.2f %.2f %.2f %.2f\n",
           scale, bad, badT, maxd, ratio, out[0], out[1], out[2], out[3], ref[0], ref[1], ref[2], ref[3]);
  }
  return 0;
}
