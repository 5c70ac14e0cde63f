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
// two floats -> one register of two bf16: the vector conversion lowers to ONE v_cvt_pk_bf16_f32 (the scalar form
// (uint)f2bf(lo) | f2bf(hi) << 16 cost two conversions and a v_perm in the attention loops)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
DEVI uint32_t pack_bf2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
DEVI float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
DEVI float bf_hi(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), the total in every lane: two quad permutes and two row
// rotations, folded into the adds as v_add_f32_dpp — no LDS traffic. (__shfl_xor compiles to ds_bpermute_b32: an LDS
// round trip and an lgkmcnt wait per step.) Fixed order: bitwise reproducible.
template <int CTRL>
DEVI float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
DEVI float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8
  return v;
}
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

// f32 -> OCP fp8, 4 values -> 4 bytes. The conversion instructions round to nearest even but do NOT saturate (1000 ->
// 0x7f = NaN in e4m3fn, inf in e5m2: tools/probe_fp8.hip), so clamp to the largest finite value first.
DEVI uint32_t pack_fp8x4(float a, float b, float c, float d, bool bf8) {
  const float mx = bf8 ? 57344.f : 448.f;
  // one v_med3_f32 per value (fminf(fmaxf()) is two instructions plus canonicalising maxima on MFMA outputs)
  a = __builtin_amdgcn_fmed3f(a, -mx, mx); b = __builtin_amdgcn_fmed3f(b, -mx, mx);
  c = __builtin_amdgcn_fmed3f(c, -mx, mx); d = __builtin_amdgcn_fmed3f(d, -mx, mx);
  int v = 0;
  if (bf8) { v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true); }
  else { v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true); }
  return (uint32_t)v;
}
// Running maximum of |x| of one fp8 site. Thousands of waves report per launch and same-address atomics retire at
// ~12 ns each (MI355X_MICROARCH.md, fanin: 8192 of them = 100 us, measured as a 2x slower LayerNorm), so a site is
// F8_SLOTS words on separate 64-byte lines and a wave reports into the slot its id selects: the atomics of a launch
// spread over 64 lines and retire in parallel; plb_launch_fp8_scales takes the maximum over the slots.
// IEEE order == unsigned order for x >= 0.
constexpr int F8_SLOTS = 64, F8_STRIDE = 16;
DEVI void atomic_max_abs(float* site, float v, unsigned int who) {
  unsigned int* dst = (unsigned int*)site + (who & (F8_SLOTS - 1)) * F8_STRIDE;
  const unsigned int bits = __float_as_uint(v);
  if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
}

// gelu_new (HF activations.py:59-66): 0.5 x (1 + tanh(z)), z = sqrt(2/pi) (x + 0.044715 x^3)
//   = x * sigmoid(2z) = x / (1 + exp2(x * (K1 + K3 x^2))),  K1 = -2 sqrt(2/pi) log2(e), K3 = 0.044715 K1.
// One v_exp + one v_rcp + 5 plain VALU per value (the GEMM epilogues evaluate it 128x per lane per tile).
// Saturates cleanly: exp2 -> inf gives rcp -> 0 (x -> -inf), exp2 -> 0 gives s = 1 (x -> +inf).
DEVI float gelu_sigmoid(float x, float x2) {
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x2, -0.10294324f, -2.3022082f));
  return __builtin_amdgcn_rcpf(1.0f + e);
}
DEVI float gelu_new_f(float x) { return x * gelu_sigmoid(x, x * x); }
// d/dx = s + x s (1 - s) d(2z)/dx,  d(2z)/dx = 2 sqrt(2/pi) (1 + 3*0.044715 x^2)
DEVI float gelu_new_grad_f(float x) {
  const float x2 = x * x;
  const float s = gelu_sigmoid(x, x2);
  const float w = x * __builtin_fmaf(x2, 0.21406444f, 1.5957691f);
  return __builtin_fmaf(s, w * (1.0f - s), s);
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements, delivered
// column-major (lane i of the group gets column i of the 4 rows). Lane 4q+p supplies the address
// of row q, columns 4p..4p+3.  Verified by tools/probe_hw.hip.
DEVI s16x4 lds_read_tr16(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}
// the same from a byte address inside the workgroup's LDS (LDS_ADDR of the image + offsets)
DEVI s16x4 lds_read_tr16_addr(uint32_t lds_byte_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lds_byte_addr);
}

// DMA in the scalar-base form (global_load_lds v_offset, s[base:base+1]): the per-lane 32-bit byte offsets are lane
// constants for the whole kernel and the tile's base address advances on the scalar unit — through the builtin hipcc
// keeps one running 64-bit pointer per instruction in VGPRs (and a 64-bit VALU add each per tile).
// M0 = LDS byte address of the 1-KiB (256-B for the dword form) destination; one wait state after writing M0.
// M0 is written and read inside ONE statement and is not on the clobber list: hipcc reserves M0 and ignores such a
// clobber (it only draws -Winline-asm). What makes this safe is that no compiler-generated instruction of these kernels
// reads M0 — plbert_amd/build.py: verify_m0 checks the disassembly of every code object for exactly that as a POST-LINK
// step of every build (a violating library is deleted, the build fails), and tests/test_cabi_and_host.py runs it again.
typedef __attribute__((address_space(3))) char lds_char;
#define LDS_ADDR(ptr) ((uint32_t)(uintptr_t)(lds_char*)(ptr))
#define DMA16(sbase, voff, ldsaddr) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
#define DMA4(sbase, voff, ldsaddr) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
#define DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous chunk of the logical tile order. Bijective for any nwg (cdna guide §5, T1).
DEVI int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
