// Shared device/host helpers for librerank_mi355 (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bit pattern
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RR_WAVE 64

// ---- bf16 <-> f32 (round-to-nearest-even; NaN stays NaN via the plain cast, which hipcc
// lowers to v_cvt_pk_bf16_f32 on gfx950 — MI355X_MICROARCH "Correctness boundaries").
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));   // one v_cvt_pk_bf16_f32
}

// ---- 16-bit operand type of the MFMA path: DT 0 = bf16 (default, what the reference's bf16-mixed runs use),
// DT 1 = fp16 (same MFMA rate, 3 more mantissa bits; buffers are the same uint16 carriers).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int DT>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (DT == 0) {
    return pack2bf(lo, hi);
  } else {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    const f32x2_ v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_));   // round-to-nearest-even
  }
}
// the two 16-bit values of a packed word back as floats (exact)
template <int DT>
__device__ __forceinline__ float2 unpack2(uint32_t u) {
  if constexpr (DT == 0) {
    return make_float2(__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u));
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    const f16x2_ v = __builtin_bit_cast(f16x2_, u);
    return make_float2((float)v[0], (float)v[1]);
  }
}
// v_fma_mix_f32 (fp32 multiply-add whose operands may be the halves of packed fp16 registers): float(h) + float(l) and
// f - float(h) in ONE instruction each instead of convert + convert + add / convert + subtract.  h * 1.0 and h * -1.0 are
// exact, so the results are bit for bit those of the separate instructions.  hipcc does not form it from C++ (it emits
// v_cvt_f32_f16 + v_add_f32).  HALF: 0 = low, 1 = high half of the packed word.  VOP3P like the packed-f32 ops of the
// stale-lane hazard (DESIGN.md "Numerics"): a caller that feeds freshly loaded registers fences them first (mix_fence).
// `tok` (mix_fence) is an ordering-only operand: it ties the instruction behind the fence of the loads it reads.
template <int HALF>
__device__ __forceinline__ float mix_add_f16(uint32_t h, uint32_t l, uint32_t tok) {
  float d;
  if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(l), "v"(tok));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(l), "v"(tok));
  return d;
}
template <int HALF>
__device__ __forceinline__ float mix_add_f16_f32(float h, uint32_t l, uint32_t tok) {      // h (fp32) + float(half of l)
  float d;
  if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(l), "v"(h), "v"(tok));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(l), "v"(h), "v"(tok));
  return d;
}
template <int HALF>
__device__ __forceinline__ float mix_sub_f16(float f, uint32_t h) {           // f - float(half of h)
  float d;
  if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(f));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(f));
  return d;
}
// Two wait states between a counted vmcnt release of freshly loaded registers and their first VOP3P reader.  The loaded
// registers are INPUTS only (as in/out operands of an asm statement hipcc copied them out of the load destinations, with a
// vmcnt(0) behind every load: the next pass's residual loads became synchronous, 4.9k -> 36k cycles per tile); the
// returned token orders the readers behind the fence.
__device__ __forceinline__ uint32_t mix_fence(const uint4& a, const uint4& b) {
  uint32_t t;
  asm volatile("s_nop 1\n\tv_mov_b32 %0, 0" : "=v"(t) : "v"(a.x), "v"(b.x));
  return t;
}
__device__ __forceinline__ uint32_t mix_fence(const uint4& a, const uint2& b) {
  uint32_t t;
  asm volatile("s_nop 1\n\tv_mov_b32 %0, 0" : "=v"(t) : "v"(a.x), "v"(b.x));
  return t;
}
// ---- streaming accesses of the GEMM epilogues.  An output row is written once and next read by another launch, a residual
// row is read once: neither is worth an L2 line, and the lines they would take are the operand panels the workgroups of an
// XCD share (RR_NT bit 0: outputs stored with the `nt` hint, bit 1: residual rows loaded with it; measured in DESIGN.md).
#ifndef RR_NT
#define RR_NT 1
#endif
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_;
__device__ __forceinline__ void store_stream(void* p, const uint4& v) {
  if constexpr ((RR_NT & 1) != 0) __builtin_nontemporal_store(u32x4_{v.x, v.y, v.z, v.w}, (u32x4_*)p);
  else *(uint4*)p = v;
}
__device__ __forceinline__ void store_stream(void* p, const uint2& v) {
  if constexpr ((RR_NT & 1) != 0) __builtin_nontemporal_store(u32x2_{v.x, v.y}, (u32x2_*)p);
  else *(uint2*)p = v;
}
__device__ __forceinline__ void store_stream(void* p, const float4& v) {
  if constexpr ((RR_NT & 1) != 0) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)p);
  else *(float4*)p = v;
}
__device__ __forceinline__ uint4 load_stream_u4(const void* p) {
  if constexpr ((RR_NT & 2) != 0) {
    const u32x4_ v = __builtin_nontemporal_load((const u32x4_*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *(const uint4*)p;
  }
}
__device__ __forceinline__ float4 load_stream_f4(const void* p) {
  if constexpr ((RR_NT & 2) != 0) {
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *(const float4*)p;
  }
}
__device__ __forceinline__ uint2 load_stream_u2(const void* p) {
  if constexpr ((RR_NT & 2) != 0) {
    const u32x2_ v = __builtin_nontemporal_load((const u32x2_*)p);
    return make_uint2(v.x, v.y);
  } else {
    return *(const uint2*)p;
  }
}
// ---- 8-bit `lo` half of the split residual stream (GemmFold::lo_bits == 8): lo = x - hi travels as OCP e5m2 ("bf8") of
// lo * 2^RR_LO8_SHIFT, converted by gfx950's scaled pack / unpack instructions (the scale operand is the MX block scale 2^-SHIFT:
// encode divides by it, decode multiplies).  |lo| <= ulp(hi) / 2, so three significant bits of lo put x at 14 (fp16 hi) / 11
// (bf16 hi) significant bits in the worst case, a power of two beyond what the 16-bit MFMA operand keeps of it anyway; the
// shift keeps lo * 2^SHIFT a NORMAL e5m2 number (>= 2^-14) for every |x| >= 2^-7 (fp16 hi) and below the e5m2 maximum (57 344)
// for |x| < 2^22 (fp16 rows end at 6.5e4; bf16 rows beyond 9e5 saturate lo, i.e. fall back to the hi half's precision); smaller
// x keep an absolute error <= 2^-21.  WORD: 0 / 1 = the low / high 16 bits of the packed dword (two e5m2 values each).
constexpr int RR_LO8_SHIFT = 4;
#ifndef RR_RESID_LO8_DEFAULT
#define RR_RESID_LO8_DEFAULT (-1)   // process-wide default of the "resid_lo8" option (rr_api.hip): -1 = by operand type (fp16: on, bf16: off)
#endif
// Memory layout of the 8-bit lo rows, private to the residual epilogues that write and read them: rows r and r + 16 of an aligned
// group of 32 rows are interleaved in units of 8 columns, so that a lane's 8 columns of BOTH rows are one 16-byte chunk (8-byte
// accesses reach 0.54 - 0.70 x of the 16-byte rate here).  Byte offset of the chunk that holds columns col .. col + 7 (col % 8 == 0)
// of row r (r % 32 < 16: first 8 bytes) and row r + 16 (last 8 bytes); ld = columns per row.  The buffer spans ceil(M / 32) * 32 rows.
__device__ __forceinline__ size_t lo8_pair_offset(int r, int col, int ld) {
  return ((size_t)(r >> 5) * 16 + (r & 15)) * (size_t)(2 * ld) + (size_t)col * 2;
}
template <int WORD>
__device__ __forceinline__ float2 lo8_decode(uint32_t w, float mx_scale, uint32_t tok) {   // two e5m2 -> fp32, times mx_scale
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  f32x2_ d;
  if constexpr (WORD == 0) asm("v_cvt_scalef32_pk_f32_bf8 %0, %1, %2" : "=v"(d) : "v"(w), "s"(mx_scale), "v"(tok));
  else asm("v_cvt_scalef32_pk_f32_bf8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(d) : "v"(w), "s"(mx_scale), "v"(tok));
  return make_float2(d.x, d.y);
}
template <int WORD>
__device__ __forceinline__ uint32_t lo8_encode(uint32_t old, float a, float b, float mx_scale) {   // e5m2(a / mx_scale), e5m2(b / mx_scale) into half WORD of `old`
  if constexpr (WORD == 0) asm("v_cvt_scalef32_pk_bf8_f32 %0, %1, %2, %3" : "+v"(old) : "v"(a), "v"(b), "s"(mx_scale));
  else asm("v_cvt_scalef32_pk_bf8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(old) : "v"(a), "v"(b), "s"(mx_scale));
  return old;
}
__device__ __forceinline__ uint32_t pack2rt(float lo, float hi, int dt) { return dt ? pack2<1>(lo, hi) : pack2<0>(lo, hi); }
template <int DT>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (DT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (DT == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// erf-GELU, gelu(x) = max(x, 0) - z Phi(-z) with z = min(|x|, 5.7) and log2 Phi(-z) a polynomial in z: plain VALU operations
// and ONE transcendental (the exp2) instead of eleven and two (exp2 + rcp) for the Abramowitz-Stegun 7.1.26 form of round 1 — the
// GELU runs exposed in the FFN-up epilogue, 128 values per lane and tile, vector-issue-bound.
// RR_GELU_DEGREE 5 (round 5): a minimax fit of the ABSOLUTE error of z 2^p(z) on [0, 5.7] (Lawson reweighting:
// tools/study/fit_gelu_tail.py): |gelu - exact| <= 6.4e-7 over [-12, 12] evaluated in fp32 with fused multiply-adds, relative error <= 2.8e-4
// wherever |gelu| >= 1e-3 — two FMAs per value fewer than the degree-7 fit of rounds 2-4 (4.8e-7 / 3.4e-5; RR_GELU_DEGREE 7 keeps
// it for A/B runs), still two to three orders below the 16-bit rounding of the output (fp16: 4.9e-4 relative).
// Beyond |x| = 5.7 the neglected term is below 6e-8.
#ifndef RR_GELU_DEGREE
#define RR_GELU_DEGREE 5
#endif
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fminf(fabsf(x), 5.7f);
#if RR_GELU_DEGREE == 7
  float p = -1.403277905e-06f;
  p = fmaf(p, z, 5.490869060e-05f);
  p = fmaf(p, z, -8.929630618e-04f);
  p = fmaf(p, z, 8.417915996e-03f);
  p = fmaf(p, z, -5.388785911e-02f);
  p = fmaf(p, z, -4.584285712e-01f);
  p = fmaf(p, z, -1.151314700e+00f);
  p = fmaf(p, z, -9.999805559e-01f);
#else
  float p = -4.733084352e-04f;
  p = fmaf(p, z, 7.084545679e-03f);
  p = fmaf(p, z, -5.182733759e-02f);
  p = fmaf(p, z, -4.599924982e-01f);
  p = fmaf(p, z, -1.150787830e+00f);
  p = fmaf(p, z, -1.000037670e+00f);
#endif
  return fmaf(-z, __builtin_amdgcn_exp2f(p), __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_huge_valf()));   // med3(x, 0, +inf) = max(x, 0)
}

// The same function on two values with the Horner chain as v_pk_fma_f32 (two fp32 lanes per instruction at the full VALU
// rate: 8 packed multiply-adds for the pair instead of 16 scalar ones; |x|, min, exp2 and med3 have no packed form).
// Every packed operand is VALU-produced (z from v_min, p from the previous packed op, coefficients from SGPR pairs), so the
// stale-lane hazard of packed-f32 ops behind a vmcnt release (DESIGN.md "Numerics") cannot arise here; each lane's result
// is the same IEEE fma chain as gelu_erf_fast, bit for bit.
#ifndef RR_PK_GELU
#define RR_PK_GELU 1
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
  const f32x2 z = {fminf(fabsf(x.x), 5.7f), fminf(fabsf(x.y), 5.7f)};
#if RR_GELU_DEGREE == 7
  f32x2 p = {-1.403277905e-06f, -1.403277905e-06f};
  p = __builtin_elementwise_fma(p, z, f32x2{5.490869060e-05f, 5.490869060e-05f});
  p = __builtin_elementwise_fma(p, z, f32x2{-8.929630618e-04f, -8.929630618e-04f});
  p = __builtin_elementwise_fma(p, z, f32x2{8.417915996e-03f, 8.417915996e-03f});
  p = __builtin_elementwise_fma(p, z, f32x2{-5.388785911e-02f, -5.388785911e-02f});
  p = __builtin_elementwise_fma(p, z, f32x2{-4.584285712e-01f, -4.584285712e-01f});
  p = __builtin_elementwise_fma(p, z, f32x2{-1.151314700e+00f, -1.151314700e+00f});
  p = __builtin_elementwise_fma(p, z, f32x2{-9.999805559e-01f, -9.999805559e-01f});
#else
  f32x2 p = {-4.733084352e-04f, -4.733084352e-04f};
  p = __builtin_elementwise_fma(p, z, f32x2{7.084545679e-03f, 7.084545679e-03f});
  p = __builtin_elementwise_fma(p, z, f32x2{-5.182733759e-02f, -5.182733759e-02f});
  p = __builtin_elementwise_fma(p, z, f32x2{-4.599924982e-01f, -4.599924982e-01f});
  p = __builtin_elementwise_fma(p, z, f32x2{-1.150787830e+00f, -1.150787830e+00f});
  p = __builtin_elementwise_fma(p, z, f32x2{-1.000037670e+00f, -1.000037670e+00f});
#endif
  const f32x2 e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  const f32x2 m = {__builtin_amdgcn_fmed3f(x.x, 0.0f, __builtin_huge_valf()), __builtin_amdgcn_fmed3f(x.y, 0.0f, __builtin_huge_valf())};
  return __builtin_elementwise_fma(-z, e, m);
}

// LDS byte offset of 16-byte chunk `c` (0..7) of row `row` in a [rows][64 x bf16] tile image
// (128-byte rows).  XOR with (row>>1)&7 keeps every ds_read_b128 lane group (16 rows of one
// chunk column, cdna_hip_programming.md §2 / T2) on 16 distinct 16-byte slots of the
// 256-byte bank row => conflict-free.
__device__ __forceinline__ int swz128(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

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
// same, global address = 64-bit scalar base + 32-bit per-lane byte offset (keeps 1 VGPR per source instead of 2)
__device__ __forceinline__ void glds16_so(const void* sbase, uint32_t voff, uint32_t lds_dst_wave_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst_wave_base)
      : "memory");
}

// 4 bytes per lane into LDS (wave-uniform destination + lane*4): used as a no-register "touch" that pulls one 128-byte
// line per lane towards the XCD's L2 ahead of the real loads (the bytes that land in LDS are never read)
__device__ __forceinline__ void glds4_so(const void* sbase, uint32_t voff, uint32_t lds_dst_wave_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst_wave_base)
      : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// host-side launcher signatures (implemented in the .hip files) ---------------------------------
enum GemmEpilogue {
  EPI_BIAS_BF16 = 0,       // C = bf16(A W^T + b)
  EPI_BIAS_GELU_BF16 = 1,  // C = bf16(gelu_erf(A W^T + b))
  EPI_BIAS_F32 = 2,        // C = f32(A W^T + b)
  EPI_BIAS_TANH_BF16 = 3,  // C = bf16(tanh(A W^T + b))
  EPI_BIAS_RESID_F32 = 4,  // C = f32(A W^T + b + R)
  EPI_BIAS_QGELU_BF16 = 5  // C = 16bit(quick_gelu(A W^T + b)), x * sigmoid(1.702 x)  (CLIP ViT MLP)
};

// A [M,Kd] bf16 (row stride lda), W [N,Kd] bf16 (row stride ldw), bias [N] f32 or null,
// resid [M,N] f32 (row stride ldr, EPI 4) ; C row stride ldc (elements of its type).
hipError_t rr_launch_gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias,
                          const float* resid, int ldr, void* C, int ldc, int M, int N, int Kd,
                          int epilogue, int dt, hipStream_t st);

// LayerNorm folded into the consumer GEMM (gemm_bf16.hip "LnResid"): producer outputs of a residual GEMM (`x16` 16-bit copy
// of the fp32 result rows, row stride ldx; `part` [M][nparts] per-128-column-group (mean, sum of squared deviations),
// nparts = ceil(N / 128)) and/or consumer inputs (`in_stats` [M] (mean, rstd) of the raw A rows, `csum` [N] column sums
// of the gamma-folded weight; `bias` then holds d = W beta + b).
struct GemmFold {
  bf16_t* x16 = nullptr;
  int ldx = 0;
  float* part = nullptr;
  int nparts = 0;
  const float* in_stats = nullptr;
  const float* csum = nullptr;
  // split residual stream (persistent ring kernel only; rr_gemm_split_ok): a pre-LayerNorm row x is carried as the pair
  // hi = 16-bit operand rounding of x (the very `x16` rows the consumer GEMM reads) + lo = fp16(x - hi), 22 (fp16 operands)
  // or 19 (bf16) significant bits, instead of a third copy in fp32: the residual epilogue moves 8 instead of 10 bytes
  // per element.  r_hi / r_lo: where THIS launch's residual rows come from (in place of `resid`); lo_out: where the lo
  // half of the output rows goes (the hi half is x16; the fp32 output C is then not written).  In-place use (r_hi == x16,
  // r_lo == lo_out) is allowed: an element is read and written by the same thread, read first.
  const bf16_t* r_hi = nullptr;
  const bf16_t* r_lo = nullptr;
  int ld16 = 0;
  bf16_t* lo_out = nullptr;
  // 16: lo rows are fp16 [M][ld16]; 8: e5m2 bytes of lo * 2^RR_LO8_SHIFT, [M][ld16] BYTES (r_lo / lo_out point at bytes): the
  // residual epilogue moves 6 instead of 8 bytes per element and touches 3 instead of 4 cache lines per 64 columns
  int lo_bits = 16;
};
// true when a GEMM with M x N output will run on the kernel that implements the split residual stream
bool rr_gemm_split_ok(int M, int N);
hipError_t rr_launch_gemm_ln(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                             int ldr, const float* ln_stats, const float* ln_gamma, const float* ln_beta, void* C, int ldc,
                             int M, int N, int Kd, int epilogue, int dt, hipStream_t st);
hipError_t rr_launch_gemm_fold(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const float* resid,
                               int ldr, const float* ln_stats, const float* ln_gamma, const float* ln_beta,
                               const GemmFold& fold, void* C, int ldc, int M, int N, int Kd, int epilogue, int dt,
                               hipStream_t st);
// (mean, sum of squared deviations) per 128-column group -> (mean, rstd) per row (Chan's pairwise merge, fp32)
// Walking direction of the next large launch of the layer chain (QKV -> attention -> attention-out -> FFN-up -> FFN-down -> ...):
// consecutive launches walk the rows in opposite directions (rr_set_tuning("m_alternate", 0) switches it off), so that a consumer
// starts on the rows its producer wrote LAST — the ones the 256 MB memory-side cache may still hold.  Speed only: every tile /
// block computes the same values whichever order they run in (logits bit-identical; three interleaved A/Bs of the c3 step on one
// box: 87.44 / 87.61 / 87.46 ms off, 87.22 / 87.50 / 87.28 on).  A process-wide launch counter decides, nothing else depends on it;
// launches of fewer than two tiles per CU / 2 048 attention blocks do not take part.  0 = ascending.
int rr_m_direction_next();
constexpr float RR_RANGE_SS_FP16 = 9.0e8f;      // row sum of squares below which no element can exceed 3e4 (fp16 operand rows)
hipError_t rr_launch_ln_finalize(const float* part, int nparts, int cols, float eps, int rows, float* stats, hipStream_t st,
                                 int* range_flag = nullptr, float range_ss = RR_RANGE_SS_FP16);

// fp8 (csrc/gemm_fp8.hip, elementwise.hip)
hipError_t rr_launch_gemm_fp8(const uint8_t* A, int lda, const uint8_t* W, int ldw, const float* bias, float scale,
                              const float* row_scale, const float* col_scale, void* C, int ldc, int M, int N, int Kd,
                              int epilogue, int dt, hipStream_t st, float out_mul = 1.0f, const float* resid = nullptr,
                              int ldr = 0, const float* rstats = nullptr, const float* rgamma = nullptr,
                              const float* rbeta = nullptr);
bool rr_gemm_fp8_ring_ok(int M, int N, int Kd);   // the shapes whose e4m3 GEMM runs on the persistent ring (epilogues 3 / 4 exist there only)
hipError_t rr_launch_layernorm_q8(const float* x, const float* gamma, const float* beta, float eps, int rows, int cols,
                                  uint8_t* out8, float* row_scale, float* stats_out, hipStream_t st);

// Multi-head attention, head dim 64.  q rows: q + (bq*Tq + t)*q_stride + head*64, where
// bq = (b + q_batch_off) / q_batch_div;  k,v rows: (b*Tk + t)*kv_stride + head*64;  key_bias [B,Tk] f32 additive
// (0 valid, -1e30 masked) or null;  out rows: (b*Tq + t)*out_stride + head*64 (bf16).
hipError_t rr_launch_attention(const bf16_t* q, int q_stride, int q_batch_div, int q_batch_off, const bf16_t* k,
                               const bf16_t* v, int kv_stride, const float* key_bias, int B,
                               int heads, int Tq, int Tk, bf16_t* out, int out_stride, int dt,
                               hipStream_t st, const float* dense_bias = nullptr, int dense_ld = 0, long schedule_blocks = 0,
                               int fixed_mode = -1);
hipError_t rr_launch_attention_segs(const bf16_t* q, int q_stride, const bf16_t* k, const bf16_t* v, int kv_stride,
                                    const float* key_bias, int heads, int nseg, const int* seg_n, const int* seg_len,
                                    const long long* seg_row0, bf16_t* out, int out_stride, int dt, hipStream_t st,
                                    long schedule_blocks, int fixed_mode = -1);
// dense_bias: optional additive bias [B][Tq][dense_ld] (dense_ld a multiple of 64 >= Tk, zero padded), PreFLMR fusion
hipError_t rr_launch_fusion_adj(const float* scores, int S, int Tq, int Tc, float mult, int pair0, int n, float* adj, int ld,
                                hipStream_t st, int row0 = 2);

// CLIP ViT front end: im2col of the stride = kernel patch convolution, and [class | patches] + position -> pre_layrnorm
hipError_t rr_launch_vit_im2col(const float* px, bf16_t* out, int B, int IS, int ps, int Kp, int dt, hipStream_t st);
hipError_t rr_launch_vit_embed_ln(const float* patches, const float* cls_emb, const float* pos, const float* gamma,
                                  const float* beta, float eps, int rows, int T, int cols, float* o32, hipStream_t st);

hipError_t rr_launch_layernorm(const float* x, const float* gamma, const float* beta, float eps,
                               int rows, int cols, float* out_f32, bf16_t* out_bf16, int dt, hipStream_t st);
