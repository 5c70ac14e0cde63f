// Shared device helpers for the gfx950 (CDNA4) kernels of the PL-BERT hot path.
// Wave = 64 lanes everywhere; bf16 travels as raw uint16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define DEVI __device__ __forceinline__

DEVI float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
DEVI bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(bf16_t, b); }
DEVI uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
DEVI float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
DEVI float bf_hi(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }

DEVI float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEVI float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// gelu_new (HF activations.py:59-66): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
DEVI float tanh_fast(float z) {
  // tanh(z) = 1 - 2 / (exp(2z) + 1); exp via exp2. Saturates cleanly for |z| large.
  float e = __builtin_amdgcn_exp2f(z * 2.885390081777927f);  // 2*log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
DEVI float gelu_new_f(float x) {
  const float c = 0.7978845608028654f;
  float t = tanh_fast(c * (x + 0.044715f * x * x * x));
  return 0.5f * x * (1.0f + t);
}
DEVI float gelu_new_grad_f(float x) {
  const float c = 0.7978845608028654f;
  float x2 = x * x;
  float t = tanh_fast(c * (x + 0.044715f * x * x2));
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 0.134145f * x2);
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements, delivered
// column-major (lane i of the group gets column i of the 4 rows). Lane 4q+p supplies the address
// of row q, columns 4p..4p+3.  Verified by tools/probe_hw.hip.
DEVI s16x4 lds_read_tr16(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous chunk of the logical tile order. Bijective for any nwg (cdna guide §5, T1).
DEVI int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
