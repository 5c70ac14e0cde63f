// Shared epilogue of the NT GEMM kernels: one lane holds 4 consecutive output columns of one row.
#pragma once
#include "common.h"
#include "plbert_kernels.h"

// v = acc (+bias) (+residual) (*gelu_new'(aux) when ACT == 2) -> bf16 C (and C2 = gelu_new(C) when
// ACT == 1) or fp32 Cf. Rows >= Mstore and columns >= N are not stored. Returns the values as stored
// (bf16-rounded for bf16 outputs; zeros when nothing was stored) so callers can form column sums.
template <int ACT, bool OUTF32>
DEVI f32x4 nt_epilogue(const PlbGemmNT& p, f32x4 v, int m, int n0) {
  if (m >= p.Mstore || n0 >= p.N) return f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    float4 b = *(const float4*)(p.bias + n0);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (p.res) {
    uint2 r = *(const uint2*)(p.res + (size_t)m * p.ldr + n0);
    v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
  }
  if (ACT == 2) {  // gelu backward: multiply by gelu_new'(u)
    uint2 u = *(const uint2*)(p.aux + (size_t)m * p.ldaux + n0);
    v[0] = v[0] != 0.f ? v[0] * gelu_new_grad_f(bf_lo(u.x)) : 0.f; v[1] = v[1] != 0.f ? v[1] * gelu_new_grad_f(bf_hi(u.x)) : 0.f;
    v[2] = v[2] != 0.f ? v[2] * gelu_new_grad_f(bf_lo(u.y)) : 0.f; v[3] = v[3] != 0.f ? v[3] * gelu_new_grad_f(bf_hi(u.y)) : 0.f;
  }
  if (OUTF32) {
    *(float4*)(p.Cf + (size_t)m * p.ldcf + n0) = make_float4(v[0], v[1], v[2], v[3]);
    return v;
  } else {
    uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
    *(uint2*)(p.C + (size_t)m * p.ldc + n0) = o;
    if (ACT == 1) {  // gelu forward: C keeps the pre-activation u (rounded to bf16, as consumed by the
                     // backward), C2 = gelu_new(u)
      uint2 g;
      g.x = pack_bf2(gelu_new_f(bf_lo(o.x)), gelu_new_f(bf_hi(o.x)));
      g.y = pack_bf2(gelu_new_f(bf_lo(o.y)), gelu_new_f(bf_hi(o.y)));
      *(uint2*)(p.C2 + (size_t)m * p.ldc2 + n0) = g;
    }
    return f32x4{bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};
  }
}
