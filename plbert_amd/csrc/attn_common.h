// Helpers shared by the attention kernels (attn.hip: forward, two-kernel backward; attn_bwd_fused.hip: the single-kernel
// backward): MFMA wrappers with the timing-build switches, the LDS image layouts and their fragment reads, the
// accumulator-as-operand conversion and the transposed store through a per-wave LDS patch.
#pragma once
#include "common.h"
#include "plbert_kernels.h"

namespace {

// Timing-only builds (tools/build_ab.sh <name> -DATTN_DBG=mask; results are wrong by construction): 1 = no exponentials,
// 2 = no MFMA, 4 = operand fragments are not read from LDS, 8 = no barriers, 16 = the kernels return at once (launch cost),
// 32 = no tiles (prologue + epilogue only), 64 = no bias-gradient column sums, 128 = no output stores. They say where a tile's time goes.
#ifndef ATTN_DBG
#define ATTN_DBG 0
#endif
#if ATTN_DBG & 2
DEVI f32x16 MFMA32(bf16x8 a, bf16x8 b, f32x16 c) { asm volatile("" ::"v"(a), "v"(b)); return c; }
#else
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
#if ATTN_DBG & 1
#define EXP2(x) (x)
#else
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#if ATTN_DBG & 8
#define TILE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define TILE_SYNC() __syncthreads()
#endif

#if ATTN_DBG & 16
#define DBG_EARLY_EXIT(p) if ((p).S > 0) return
#else
#define DBG_EARLY_EXIT(p)
#endif
#if ATTN_DBG & 32
#define DBG_TILES(n) ((n) > 100000 ? 1 : 0)
#else
#define DBG_TILES(n) (n)
#endif

constexpr float LOG2E = 1.4426950408889634f;

// [64 rows][64 cols] bf16, 128-B rows, chunk index XORed with (row>>1)&7: conflict-free ds_read_b128
// for 32 consecutive rows at one chunk.
DEVI int row_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
// [64 rows][64 cols] bf16 in [row/4][col/32][4][32] sub-tiles of 256 B: conflict-free tr reads.
DEVI int tr_off(int row, int col) { return (((row >> 2) << 1) + (col >> 5)) * 128 + (row & 3) * 32 + (col & 31); }

// A-operand fragment of X^T (X stored [row][col] in the tr layout): lane (m = cb*32 + (l&31), half h)
// element j <- X[rb*32 + 16s + 8(j>>2) + 4h + (j&3)][m]
DEVI bf16x8 tr_frag(const bf16_t* tile, int rb, int s, int cb, int lane) {
#if ATTN_DBG & 4
  bf16x8 f; asm volatile("; frag" : "=v"(f)); return f;
#endif
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3, h = g >> 1;
  const int r0 = rb * 32 + 16 * s + 4 * h + q4;
  const int c = cb * 32 + 16 * (g & 1) + 4 * p4;
  s16x4 a = lds_read_tr16(&tile[tr_off(r0, c)]);
  s16x4 b = lds_read_tr16(&tile[tr_off(r0 + 8, c)]);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// Row-layout fragment: lane (row = rb*32 + (l&31), k = ks*16 + 8h + j)
DEVI bf16x8 row_frag(const bf16_t* tile, int rb, int ks, int lane) {
#if ATTN_DBG & 4
  bf16x8 f; asm volatile("; frag" : "=v"(f)); return f;
#endif
  const int row = rb * 32 + (lane & 31);
  return *(const bf16x8*)&tile[row_off(row, ks * 2 + (lane >> 5))];
}
// The four lane-constant element offsets of row_frag(tile, 0, ks, lane), ks = 0..3 (row block rb adds rb*2048): the XOR
// swizzle keeps them from being one base + immediates, so the kernels compute them once instead of per tile.
DEVI void row_frag_offsets(int lane, int (&off)[4]) {
  const int row = lane & 31;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) off[ks] = row_off(row, ks * 2 + (lane >> 5));
}
#if ATTN_DBG & 4
#define ROW_FRAG(tile, rb, ks, off) row_frag(tile, rb, ks, 0)
#else
#define ROW_FRAG(tile, rb, ks, off) (*(const bf16x8*)&(tile)[(off)[ks] + (rb) * 2048])
#endif
// registers 8s..8s+7 of a 32x32 accumulator -> bf16x8 operand fragment (k-step s)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
DEVI bf16x8 acc_frag(const f32x16& x, int s) {
  const u32x4_t u = {pack_bf2(x[8 * s + 0], x[8 * s + 1]), pack_bf2(x[8 * s + 2], x[8 * s + 3]),
                     pack_bf2(x[8 * s + 4], x[8 * s + 5]), pack_bf2(x[8 * s + 6], x[8 * s + 7])};
  return __builtin_bit_cast(bf16x8, u);  // four packed registers ARE the fragment: no lane or byte shuffles
}
DEVI f32x16 splat16(float v) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = v;
  return z;
}
DEVI f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Store a wave's transposed accumulator pair X^T[64 c][32 r] (lane = r, registers = c) as rows
// out[r][0..63] (bf16) through a per-wave LDS patch with 144-B rows, then 16-B coalesced stores.
// colsum (optional): this wave's 64 column sums of the rows it stored (the values as rounded to bf16) — lane c sums
// column c of the patch. The engine adds these few partial rows up into the Q/K/V bias gradient instead of re-reading
// the stacked [L*T, 3H] gradient (906 MB per step at config A).
// NT: the rows leave with the non-temporal hint (ATTN_OUT_NT picks the kernels: 1 forward, 2 dQ, 4 dK / dV, 8 fused).
#ifndef ATTN_OUT_NT
#define ATTN_OUT_NT 0
#endif
// Optional fp8 copy of the rows a wave stores (fp8 mode): the values AS ROUNDED TO bf16 times qs, saturated, e4m3 or e5m2;
// amax = running max |value| of the lane (the caller reduces it and reports it to the site). g8 == nullptr: off.
// BF8: the image's format is fixed by the kernel (forward: e4m3 context, backward: e5m2 gradients) — a compile-time choice,
// so that only one conversion instruction per pair is emitted.
struct Out8 { uint8_t* g8; int ld8; float qs; bool bf8; };
template <bool NT = false, bool BF8 = false>
DEVI void store_transposed(const f32x16& a0, const f32x16& a1, float mult, bf16_t* patch, bf16_t* gout, int ldo,
                           int rows_valid, int lane, float* colsum = nullptr, bool accumulate = false,
                           const Out8* o8 = nullptr, float* amax = nullptr) {
  constexpr int PS = 72;  // elements per patch row (144 B)
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const f32x16& a = cb ? a1 : a0;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      uint2 v;
      v.x = pack_bf2(a[4 * rg + 0] * mult, a[4 * rg + 1] * mult);
      v.y = pack_bf2(a[4 * rg + 2] * mult, a[4 * rg + 3] * mult);
      *(uint2*)&patch[r * PS + cb * 32 + 8 * rg + 4 * h] = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the patch is private to this wave
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = lane + 64 * i, row = id >> 3, c = id & 7;
    uint4 v = *(const uint4*)&patch[row * PS + c * 8];
    if (row < rows_valid && (!(ATTN_DBG & 128) || v.x == 0x12345678u)) {
      typedef unsigned int u32x4nt_ __attribute__((ext_vector_type(4)));
      if (gout) {
        if constexpr (NT) __builtin_nontemporal_store(u32x4nt_{v.x, v.y, v.z, v.w}, (u32x4nt_*)(gout + (size_t)row * ldo + c * 8));
        else *(uint4*)(gout + (size_t)row * ldo + c * 8) = v;
      }
      if (o8 && o8->g8) {
        const float f0 = bf_lo(v.x), f1 = bf_hi(v.x), f2 = bf_lo(v.y), f3 = bf_hi(v.y);
        const float f4 = bf_lo(v.z), f5 = bf_hi(v.z), f6 = bf_lo(v.w), f7 = bf_hi(v.w);
        const float q = o8->qs;
        uint2 w;
        w.x = pack_fp8x4(f0 * q, f1 * q, f2 * q, f3 * q, BF8);
        w.y = pack_fp8x4(f4 * q, f5 * q, f6 * q, f7 * q, BF8);
        *(uint2*)(o8->g8 + (size_t)row * o8->ld8 + c * 8) = w;
        *amax = fmaxf(*amax, fmaxf(fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3))),
                                   fmaxf(fmaxf(fabsf(f4), fabsf(f5)), fmaxf(fabsf(f6), fabsf(f7)))));
      }
    }
  }
  if (colsum && !(ATTN_DBG & 64)) {
    // lane (rg, cg) sums columns 4 cg .. 4 cg + 3 over rows 8 rg .. 8 rg + 7 (8 reads of 8 bytes instead of one 2-byte
    // read per row and lane), the four row groups meet through two cross-row shuffles, lanes 0..15 store 16 bytes each
    const int rg = lane >> 4, cg = lane & 15;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = rg * 8 + i;
      const uint2 v = *(const uint2*)&patch[rr * PS + cg * 4];
      if (rr < rows_valid) { s0 += bf_lo(v.x); s1 += bf_hi(v.x); s2 += bf_lo(v.y); s3 += bf_hi(v.y); }
    }
    s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); s3 += __shfl_xor(s3, 16, 64);
    s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); s3 += __shfl_xor(s3, 32, 64);
    if (lane < 16) {
      float4 o = make_float4(s0, s1, s2, s3);
      if (accumulate) { const float4 q = *(const float4*)(colsum + cg * 4); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
      *(float4*)(colsum + cg * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------- staging
// K / V (forward, dQ) and Q / dO (dK,dV) tiles reach LDS by LDS-DMA straight from global memory (global_load_lds, 1 KiB
// per wave instruction): no staging registers, no ds_write, no per-tile address arithmetic. An image's swizzle is applied
// to the per-lane SOURCE address; wave w writes rows [16w, 16w+16) of a 64-row image, 8 rows per instruction:
//  row image (row_off): LDS (row, chunk') <- source chunk chunk' ^ ((row >> 1) & 7)
//  tr image  (tr_off):  LDS 16-byte unit u of 256-byte sub-tile t <- source (row 4(t>>1) + (u>>2), col 32(t&1) + 8(u&3))
// Two stages: the DMA of tile t+1 is issued when tile t starts and waited for (vmcnt) at the barrier that ends it.
// (DMA16 / DMA4 / LDS_ADDR / DMA_WAIT: common.h)
// lane constants of the staging: image rows / source columns (elements) of this lane's two instructions
struct StageLane { int rA, cA0, cA1, rT, cT; };
DEVI StageLane stage_lane(int wave, int lane) {
  StageLane g;
  g.rA = wave * 16 + (lane >> 3);
  g.cA0 = ((lane & 7) ^ (lane >> 4)) * 8;
  g.cA1 = ((lane & 7) ^ (4 + (lane >> 4))) * 8;
  g.rT = wave * 16 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  g.cT = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  return g;
}

}  // namespace
