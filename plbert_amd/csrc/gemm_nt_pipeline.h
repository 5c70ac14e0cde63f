// The multi-phase, counted-vmcnt LDS-DMA pipeline kernel of the large NT GEMMs (gfx950), shared by the bf16 build
// (gemm_big.hip) and the fp8 build (gemm_fp8.hip). See gemm_big.hip for the description of the pipeline.
#pragma once
#include "common.h"
#include "plbert_kernels.h"
#include "gemm_epilogue.h"

#ifndef NT_DBG
#define NT_DBG 0
#endif
// bf16 output tiles leave as full lines, those of the gelu and LayerNorm forms with the non-temporal hint: 25-134 MB of
// output per launch otherwise sweep the weights and the activation panels out of the XCD's 4 MB L2 (PMC: the 2-round gelu
// launches fetched their weights once per round). Measured per form on one box (profiles/r03_nt_store_ab.txt): gelu forms
// -0.15 ms/step, gelu + LayerNorm forms -0.29 ms/step; on the plain launches the hint is neutral (the QKV output is what
// the attention kernel reads next: it lost what the GEMM won). NT_OUT_NT (bit mask, A/B builds): 1 plain launches (act 0),
// 2 gelu forms (1, 2, 7, 8), 4 LayerNorm forms (5, 6), 8 only the pre-LayerNorm image of form 5.
#ifndef NT_OUT_NT
#define NT_OUT_NT 6
#endif
// NT_LN_STAGE (A/B builds; default on): the LayerNorm forms bring their epilogue operand tiles (residual; form 6 also the
// pre-LayerNorm sums) into LDS by LDS-DMA as FULL lines, in the output image's layout, instead of loading them in MFMA
// layout (where one load instruction is 16 partial lines: 48 such loads cost the texture addresser ~6 us per operand).
#ifndef NT_LN_STAGE
#define NT_LN_STAGE 1
#endif
typedef unsigned int u32x4nt __attribute__((ext_vector_type(4)));
#define OUT_STORE(ptr, v)                                                                               \
  do {                                                                                                  \
    if constexpr (((NT_OUT_NT & 1) && ACT == 0) || ((NT_OUT_NT & 2) && (ACT == 1 || ACT == 2 || ACT == 7 || ACT == 8)) || \
                  ((NT_OUT_NT & 4) && (ACT == 5 || ACT == 6)))                                          \
      __builtin_nontemporal_store(u32x4nt{(v).x, (v).y, (v).z, (v).w}, (u32x4nt*)(ptr));                \
    else                                                                                                \
      *(uint4*)(ptr) = (v);                                                                             \
  } while (0)

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x8v __attribute__((ext_vector_type(8)));
// One block-scaled fp8 MFMA, K = 128: D[n][m] += sum_k B[n][k] A[m][k]; first operand (weights) e4m3, second
// (activations / gradients) e4m3 or e5m2; E8M0 scale 127 = 2^0 in every byte: unscaled fp8 arithmetic.
template <bool ABF8>
DEVI f32x4 mfma_fp8(const i32x8v& bv, const i32x8v& av, const f32x4& c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, c, 0, ABF8 ? 1 : 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// A fragment of the fp8 MFMA is 8 registers = the two 16-byte LDS reads of the bf16 kernel side by side: the halves
// are written in place (sub-registers of the tuple), so no copy is needed to form the operand.
template <int HALF>
DEVI void set_half(i32x8v& v, const i32x4v t) {
  v[4 * HALF + 0] = t[0]; v[4 * HALF + 1] = t[1]; v[4 * HALF + 2] = t[2]; v[4 * HALF + 3] = t[3];
}

namespace {

constexpr int HT = 128 * 64;  // elements per half-tile (16 KiB)
constexpr int RING = 10;      // half-tile slots of the LDS ring used by the NT kernel (160 KiB)
DEVI int ring_slot(int x) { return x >= 2 * RING ? x - 2 * RING : (x >= RING ? x - RING : x); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Retire this wave's LDS reads, then barrier: a DMA issued by another wave after the barrier may
// overwrite what those reads were fetching. The "memory" clobber keeps ds_read / global_load_lds on
// their side of the barrier; sched_barrier pins the register-only MFMAs as well.
// The wait is the builtin (0xC07F = lgkmcnt(0), vmcnt / expcnt untouched) so hipcc's own waitcnt tracking
// knows the LDS reads have retired; written as asm it would re-wait with lgkmcnt(0) at the first MFMA that
// uses those fragments, which would also stall on the reads just issued for the next phase.
#define BARRIER()                                        \
  do {                                                   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                  \
    if (!(NT_DBG & 8)) asm volatile("s_barrier" ::: "memory"); \
  } while (0)
#define PIN() __builtin_amdgcn_sched_barrier(0)

// V selects the tile: 2 -> 256x256 (2 A halves, 2 B halves), 3 -> 128x384 (1 A, 3 B), 1 -> 128x256 (1 A, 2 B;
// three half-tiles per K-tile, two phases, two half-tiles in flight: for shapes where the larger tiles
// would leave CUs idle, e.g. N = 1024 at M = 8192).
// FP8: operands are 1-byte e4m3 (A: e5m2 when ABF8 — gradients) images of the same byte geometry: a K-tile is still
// 128 bytes of every row = 128 k instead of 64, the LDS images, the DMA schedule and the fragment reads are the very
// same instructions, and one block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (unit E8M0 scales: plain fp8 arithmetic at
// twice the bf16 rate, MI355X_MICROARCH.md § Matrix cores) replaces the two bf16 MFMAs of a fragment pair. Any
// assignment of k to (lane >> 4, byte) is valid as long as A and B use the same one (tools/probe_fp8.hip), so a lane
// keeps the bf16 kernel's chunks {fq, 4 + fq} of a row: the conflict-free image needs no change. The accumulators are
// multiplied by the two per-tensor dequantisation factors in the epilogue.
// (A persistent form — 256 workgroups walking the tiles of a multi-round launch, stores left to drain under the next
// tile's K loop — was built and measured in round 3: 64.9 vs 65.0 us on the 3-round QKV launch, 71.7 vs 72.8 / 76.2 vs
// 77.1 us on the gelu forms. The hardware already turns workgroups over without waiting for their stores; what a round
// costs beside its K loop is inside the workgroup: tools/nt_stamps.py, DESIGN.md section 6. Not kept.)
// NOCS: the launch has no column-sum output (p.colpart == nullptr, checked by the launcher): the epilogue's per-value
// re-expansion of the stored bf16 and its running column sums (a third of the plain epilogue's arithmetic) are compiled out.
template <int V, int ACT, bool OUTF32, bool PF, bool FP8 = false, bool ABF8 = false, bool NOCS = false>
__global__ __launch_bounds__(512) void gemm_nt_big_kernel(PlbGemmNT p) {
  constexpr int NAH = (V == 2) ? 2 : 1;
  constexpr int NBH = (V == 3) ? 3 : 2;
  constexpr int TM = NAH * 128, TN = NBH * 128;
  constexpr int NH = NAH + NBH;  // half-tiles per K-tile, consumed in the order A0 B0 B1 [A1 | B2]
  __shared__ __attribute__((aligned(16))) bf16_t smem[RING * HT];  // 160 KiB: the whole LDS of a CU
  int tid = threadIdx.x, lane = tid & 63;   // (re-derived after the K loop in the fp8 build: see the epilogue)
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = uw >> 2, wn = uw & 3;
  const int nbn = p.N / TN;
  int bm, bn;
  if constexpr (ACT == 5 || ACT == 6) {
    // LayerNorm forms: the nbn column tiles of a row block exchange row partials inside the launch, so they sit on
    // consecutive dispatch slots of ONE XCD (block ids b, b + 8, ...: placement is a speed / latency matter only, the
    // hand-off is agent-scope). (M / 128) % 8 == 0 is checked by the launcher.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, gpx = (p.M / TM) >> 3;
    bm = xcd * gpx + slot / nbn; bn = slot % nbn;
  } else {
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    bm = logical / nbn; bn = logical % nbn;
  }
  constexpr int EPK = FP8 ? 128 : 64;  // elements of k per 128-byte K-tile row
  const int nk = p.K / EPK;
#if NT_DBG & 16
  const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if NT_DBG & 32
  unsigned long long st_[6];
  st_[0] = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- staging: each wave DMAs rows [(2w+j)*8, +8) of a half-tile, j = 0,1 (1 KiB per instruction);
  // the XOR swizzle of the image is applied to the per-lane SOURCE chunk
  const int drow = lane >> 3;
  // global addresses in BYTES: lda / ldb count elements (2 bytes, or 1 in FP8 mode)
  constexpr int EB = FP8 ? 1 : 2;
  const int ch0 = ((lane & 7) ^ ((lane >> 4) & 7)) * 16;
  const int ch1 = ((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 16;
  // DMA in the scalar-base form (common.h): the lane offsets are constants of the kernel, the (half-tile, K-tile) base
  // is formed on the scalar unit — through the builtin every instruction carried a 64-bit VGPR pointer and a VALU add
  const uint32_t vA0 = (uint32_t)(((2 * uw) * 8 + drow) * p.lda * EB + ch0), vA1 = (uint32_t)(((2 * uw + 1) * 8 + drow) * p.lda * EB + ch1);
  const uint32_t vB0 = (uint32_t)(((2 * uw) * 8 + drow) * p.ldb * EB + ch0), vB1 = (uint32_t)(((2 * uw + 1) * 8 + drow) * p.ldb * EB + ch1);
  const char* const sA_ = (const char*)p.A + (size_t)(bm * TM) * p.lda * EB;
  const char* const sB_ = (const char*)p.B + (size_t)(bn * TN) * p.ldb * EB;
  const size_t hA = (size_t)128 * p.lda * EB, hB = (size_t)128 * p.ldb * EB;
  const int dst0 = (2 * uw) * 8 * 64, dst1 = (2 * uw + 1) * 8 * 64;
  const uint32_t ldsb_ = LDS_ADDR(&smem[0]);
#define STAGE_A(slot, h, kt)                                                                                  \
  do {                                                                                                        \
    const char* b_ = sA_ + (h) * hA + (size_t)(kt) * 128;                                                     \
    DMA16(b_, vA0, ldsb_ + 2u * ((slot) * HT + dst0));                                                        \
    DMA16(b_, vA1, ldsb_ + 2u * ((slot) * HT + dst1));                                                        \
  } while (0)
#define STAGE_B(slot, h, kt)                                                                                  \
  do {                                                                                                        \
    const char* b_ = sB_ + (h) * hB + (size_t)(kt) * 128;                                                     \
    DMA16(b_, vB0, ldsb_ + 2u * ((slot) * HT + dst0));                                                        \
    DMA16(b_, vB1, ldsb_ + 2u * ((slot) * HT + dst1));                                                        \
  } while (0)
// half-tile i of a K-tile, in consumption order: A0, B0, B1, then A1 (256x256) or B2 (128x384)
#define STAGE_I(i, slot, kt)                                              \
  do {                                                                    \
    if (V == 2 && (i) == 3) STAGE_A(slot, 1, kt);                         \
    else if ((i) == 0) STAGE_A(slot, 0, kt);                              \
    else STAGE_B(slot, (i) - 1, kt);                                      \
  } while (0)
// issue half-tile number NH*t + c of the stream (c is a literal), if it exists
#define ISSUE(c)                                                          \
  do {                                                                    \
    if ((!(NT_DBG & 4) || PRO) && NH * t + (c) < htot) STAGE_I((c) % NH, ring_slot(rb + (c)), t + (c) / NH); \
  } while (0)

  // ---- fragment reads: row-in-half = wm*64 + mi*16 + frow (A) / wn*32 + ni*16 + frow (B);
  // stored chunk = (kk*4 + fq) ^ ((frow>>1)&7)
  int frow = lane & 15, fq = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int offA = (wm * 64 + frow) * 64, offB = (wn * 32 + frow) * 64;
  const int c0 = ((0 + fq) ^ fsw) << 3, c1 = ((4 + fq) ^ fsw) << 3;
  // two A and two B fragment buffers (indices are literals everywhere: plain registers). The prefetching
  // loop ping-pongs them; the staggered loop uses afr[0] and both B buffers.
  bf16x8 afr[2][4][2], bfr[2][2][2];
  i32x8v afr8[2][4], bfr8[2][2];  // FP8 mode: the same fragments as 8-register tuples (afr / bfr are then unused)
  if (NT_DBG & 2) {
    __builtin_memset(afr, 0, sizeof(afr));
    __builtin_memset(bfr, 0, sizeof(bfr));
  }
#define READ_A(ab, slot)                                                             \
  do {                                                                               \
    const bf16_t* s_ = &smem[(slot) * HT + offA];                                    \
    if constexpr (FP8) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {            \
      set_half<0>(afr8[ab][mi], *(const i32x4v*)&s_[mi * 16 * 64 + c0]);             \
      set_half<1>(afr8[ab][mi], *(const i32x4v*)&s_[mi * 16 * 64 + c1]);             \
    } else if (!(NT_DBG & 2)) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {     \
      afr[ab][mi][0] = *(const bf16x8*)&s_[mi * 16 * 64 + c0];                       \
      afr[ab][mi][1] = *(const bf16x8*)&s_[mi * 16 * 64 + c1];                       \
    }                                                                                \
  } while (0)
#define READ_B(bb, slot)                                                             \
  do {                                                                               \
    const bf16_t* s_ = &smem[(slot) * HT + offB];                                    \
    if constexpr (FP8) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {            \
      set_half<0>(bfr8[bb][ni], *(const i32x4v*)&s_[ni * 16 * 64 + c0]);             \
      set_half<1>(bfr8[bb][ni], *(const i32x4v*)&s_[ni * 16 * 64 + c1]);             \
    } else if (!(NT_DBG & 2)) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {     \
      bfr[bb][ni][0] = *(const bf16x8*)&s_[ni * 16 * 64 + c0];                       \
      bfr[bb][ni][1] = *(const bf16x8*)&s_[ni * 16 * 64 + c1];                       \
    }                                                                                \
  } while (0)

  f32x4 acc[NAH][4][NBH][2];  // [mh][mi][nh][ni]
#pragma unroll
  for (int a = 0; a < NAH; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < NBH; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (ACT == 6 && !NT_LN_STAGE) {
    // LayerNorm-backward form WITHOUT operand staging (A/B builds): the residual is the accumulators' START value (its segments arrive under the prologue's
    // DMA), not an epilogue operand — beside the kept pre segments it would not fit the epilogue's registers
    if (p.res) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const uint2 r = *(const uint2*)(p.res + (size_t)(bm * TM + wm * 64 + mi * 16 + frow) * p.ldr + bn * TN + wn * 32 +
                                            fq * 4 + nh * 128 + ni * 16);
            acc[0][mi][nh][ni] = f32x4{bf_lo(r.x), bf_hi(r.x), bf_lo(r.y), bf_hi(r.y)};
          }
    }
  }
  // swapped MFMA operands: D[row = n][col = m] -> each lane owns 4 consecutive n of one row m
#define MFMA_Q(mh, nh, ab, bb)                                                                                 \
  do {                                                                                                         \
    if constexpr (FP8) {                                                                                       \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                         \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                       \
          acc[mh][mi][nh][ni] = mfma_fp8<ABF8>(bfr8[bb][ni], afr8[ab][mi], acc[mh][mi][nh][ni]);               \
    } else if (!(NT_DBG & 1)) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                 \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                         \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                       \
          acc[mh][mi][nh][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[bb][ni][kk], afr[ab][mi][kk],     \
                                                                        acc[mh][mi][nh][ni], 0, 0, 0);         \
  } while (0)
#define MFMA_PART(mh, nh, bb)            \
  do {                                   \
    BARRIER(); PIN();                    \
    __builtin_amdgcn_s_setprio(1);       \
    MFMA_Q(mh, nh, 0, bb);               \
    __builtin_amdgcn_s_setprio(0);       \
    PIN(); BARRIER(); PIN();             \
  } while (0)
// prefetching form of a phase: one barrier, then the fragment reads of the NEXT phase are issued ahead
// of this phase's 16 MFMAs and complete underneath them (retired by the next BARRIER's lgkmcnt(0))
#define PHASE(reads, mh, nh, ab, bb)     \
  do {                                   \
    BARRIER(); PIN();                    \
    reads;                               \
    __builtin_amdgcn_s_setprio(1);       \
    MFMA_Q(mh, nh, ab, bb);              \
    __builtin_amdgcn_s_setprio(0);       \
    PIN();                               \
  } while (0)
// Wait until everything but the newest N DMA instructions has landed (N = 2 x half-tiles allowed in
// flight); once the stream has run out (an expected issue was skipped) drain completely.
#define LANDED(more, N)                                                        \
  do {                                                                         \
    if (more) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");            \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
  } while (0)

  // ---- The half-tile stream. Half-tile number h = NH*t + i (K-tile t, i-th in consumption order)
  // lives in ring slot h % 10. A slot may be re-filled once the phase that read its previous occupant
  // h - 10 has finished, so at the start of phase P the stream may advance to (last half-tile read
  // before P) + 10. That keeps 5 (128x384), 6 (256x256, 128x256) half-tiles = 80-96 KiB in flight at
  // the wait that precedes a K-tile, against 3 with a double-buffered 128 KiB layout.
  const int htot = NH * nk;
  int rb = 0;  // (NH * t) % RING
  bool PRO = true;  // NT_DBG only
  {
    const int t = 0;
    ISSUE(0); ISSUE(1); ISSUE(2); ISSUE(3); ISSUE(4); ISSUE(5); ISSUE(6); ISSUE(7); ISSUE(8);
    if constexpr (V == 2) {
      ISSUE(9);
      LANDED(htot > 9, 12);   // K-tile 0 = half-tiles 0..3
    } else if constexpr (V == 3) {
      LANDED(htot > 8, 10);   // half-tiles 0..3
    } else {
      LANDED(htot > 8, 12);   // half-tiles 0..2
    }
  }
  BARRIER();
  PRO = false;
#if NT_DBG & 32
  st_[1] = __builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (PF) {
    // ---- interleaved loop (default). Measured on the staggered loop (tools/build_dbg.sh): MFMA, fragment
    // reads and DMA cost 0.64 + 0.41 + 0.29 us per K-tile and the K-tile takes their SUM — a wave issues in
    // order, so a burst of 4-12 ds_read_b128 ahead of its MFMAs holds them back until the LDS has
    // accepted every wave's burst. Here a phase is {BARRIER, 16 MFMAs}, and the fragment reads for the
    // NEXT phase and the DMA issues are dealt one per MFMA into the shadows of those MFMAs (the guide's
    // rule: <= 2 ds_read_b128 per MFMA gap are free). sched_barrier pins that order. Fragment buffers
    // ping-pong, so the K loop is written for two K-tiles. One barrier per phase, no stagger.
    //  RAW: the vmcnt wait for K-tile t+1 ends the phase before the one whose shadows first read it, so a
    //       BARRIER lies between every wave's wait and any wave's read; it leaves only half-tiles beyond
    //       K-tile t+1 in flight, so the later phases of that K-tile need no wait of their own.
    //  WAR: reads issued in phase P retire at BARRIER(P+1) (lgkmcnt(0)); the slot is re-filled by a DMA
    //       issued after BARRIER(P+2) at the earliest.
    //  The last K-tile's "next" reads fetch stale slots (valid LDS addresses, values unused).
    READ_A(0, 0);
    READ_B(0, 1);
    int t = 0;
#define NEXT_T() do { ++t; rb += NH; rb = rb >= RING ? rb - RING : rb; } while (0)
// one MFMA of the 16 of a phase: j -> (kk, mi, ni)
#define MF(mh, nh, ab, bb, j)                                                                          \
  do {                                                                                                 \
    if constexpr (FP8) {  /* 8 MFMAs of twice the length: slot j even -> fragment pair (j >> 1) */      \
      if (((j) & 1) == 0)                                                                              \
        acc[mh][((j) >> 2) & 3][nh][((j) >> 1) & 1] = mfma_fp8<ABF8>(                                  \
            bfr8[bb][((j) >> 1) & 1], afr8[ab][((j) >> 2) & 3], acc[mh][((j) >> 2) & 3][nh][((j) >> 1) & 1]); \
    } else if (!(NT_DBG & 1))                                                                          \
      acc[mh][((j) >> 1) & 3][nh][(j) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                  \
          bfr[bb][(j) & 1][(j) >> 3], afr[ab][((j) >> 1) & 3][(j) >> 3], acc[mh][((j) >> 1) & 3][nh][(j) & 1], 0, 0, 0); \
    PIN();                                                                                             \
  } while (0)
// one ds_read_b128 of an A half (i = 0..7) / a B half (i = 0..3)
#define RA(ab, slot, i)                                                                                \
  do {                                                                                                 \
    if constexpr (FP8)                                                                                 \
      set_half<(i) & 1>(afr8[ab][(i) >> 1], *(const i32x4v*)&smem[(slot) * HT + offA + ((i) >> 1) * 16 * 64 + (((i) & 1) ? c1 : c0)]); \
    else if (!(NT_DBG & 2))                                                                            \
      afr[ab][(i) >> 1][(i) & 1] = *(const bf16x8*)&smem[(slot) * HT + offA + ((i) >> 1) * 16 * 64 + (((i) & 1) ? c1 : c0)]; \
    PIN();                                                                                             \
  } while (0)
#define RB(bb, slot, i)                                                                                \
  do {                                                                                                 \
    if constexpr (FP8)                                                                                 \
      set_half<(i) & 1>(bfr8[bb][(i) >> 1], *(const i32x4v*)&smem[(slot) * HT + offB + ((i) >> 1) * 16 * 64 + (((i) & 1) ? c1 : c0)]); \
    else if (!(NT_DBG & 2))                                                                            \
      bfr[bb][(i) >> 1][(i) & 1] = *(const bf16x8*)&smem[(slot) * HT + offB + ((i) >> 1) * 16 * 64 + (((i) & 1) ? c1 : c0)]; \
    PIN();                                                                                             \
  } while (0)
#define IS(c) do { ISSUE(c); PIN(); } while (0)
// a phase: barrier, then MFMA j followed by shadow statement sj
#define PH(mh, nh, ab, bb, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)      \
  do {                                                                                                 \
    BARRIER(); PIN();                                                                                  \
    MF(mh, nh, ab, bb, 0); s0; MF(mh, nh, ab, bb, 1); s1; MF(mh, nh, ab, bb, 2); s2; MF(mh, nh, ab, bb, 3); s3;       \
    MF(mh, nh, ab, bb, 4); s4; MF(mh, nh, ab, bb, 5); s5; MF(mh, nh, ab, bb, 6); s6; MF(mh, nh, ab, bb, 7); s7;       \
    MF(mh, nh, ab, bb, 8); s8; MF(mh, nh, ab, bb, 9); s9; MF(mh, nh, ab, bb, 10); s10; MF(mh, nh, ab, bb, 11); s11;   \
    MF(mh, nh, ab, bb, 12); s12; MF(mh, nh, ab, bb, 13); s13; MF(mh, nh, ab, bb, 14); s14; MF(mh, nh, ab, bb, 15); s15; \
  } while (0)
#define NOP_ (void)0
    // 128x256 (A B0 | B1): ph1 reads B1(t), ph2 reads A(t+1), B0(t+1)
#define STEP1(ab)                                                                                       \
  do {                                                                                                  \
    const int sB1 = ring_slot(rb + 2), sA = ring_slot(rb + 3), sB0 = ring_slot(rb + 4);                 \
    PH(0, 0, ab, 0, RB(1, sB1, 0), NOP_, RB(1, sB1, 1), NOP_, RB(1, sB1, 2), NOP_, RB(1, sB1, 3), NOP_,  \
       IS(9), NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_);                                                \
    LANDED(NH * t + 9 < htot, 8);                                                                       \
    PH(0, 1, ab, 1, RA(1 - ab, sA, 0), RA(1 - ab, sA, 1), RA(1 - ab, sA, 2), RA(1 - ab, sA, 3),         \
       RA(1 - ab, sA, 4), RA(1 - ab, sA, 5), RA(1 - ab, sA, 6), RA(1 - ab, sA, 7),                      \
       RB(0, sB0, 0), RB(0, sB0, 1), RB(0, sB0, 2), RB(0, sB0, 3), IS(10), NOP_, IS(11), NOP_);         \
  } while (0)
    // 256x256 (A0 B0 | B1 | A1 | -): ph1 reads B1(t), ph2 A1(t), ph4 A0(t+1), B0(t+1)
#define STEP2(bb)                                                                                       \
  do {                                                                                                  \
    const int sB1 = ring_slot(rb + 2), sA1 = ring_slot(rb + 3), sA0 = ring_slot(rb + 4), sB0 = ring_slot(rb + 5); \
    PH(0, 0, 0, bb, RB(1 - bb, sB1, 0), NOP_, RB(1 - bb, sB1, 1), NOP_, RB(1 - bb, sB1, 2), NOP_,        \
       RB(1 - bb, sB1, 3), NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_);                       \
    PH(0, 1, 0, 1 - bb, RA(1, sA1, 0), RA(1, sA1, 1), RA(1, sA1, 2), RA(1, sA1, 3), RA(1, sA1, 4),      \
       RA(1, sA1, 5), RA(1, sA1, 6), RA(1, sA1, 7), IS(10), NOP_, NOP_, IS(11), NOP_, NOP_, NOP_, NOP_); \
    PH(1, 1, 1, 1 - bb, NOP_, NOP_, IS(12), NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, \
       NOP_, NOP_, NOP_);                                                                               \
    LANDED(NH * t + 12 < htot, 10);                                                                     \
    PH(1, 0, 1, bb, RA(0, sA0, 0), RA(0, sA0, 1), RA(0, sA0, 2), RA(0, sA0, 3), RA(0, sA0, 4),          \
       RA(0, sA0, 5), RA(0, sA0, 6), RA(0, sA0, 7), RB(1 - bb, sB0, 0), RB(1 - bb, sB0, 1),             \
       RB(1 - bb, sB0, 2), RB(1 - bb, sB0, 3), IS(13), NOP_, NOP_, NOP_);                               \
  } while (0)
    // 128x384 (A B0 | B1 | B2): ph1 reads B1(t), ph2 B2(t), ph3 A(t+1), B0(t+1)
#define STEP3(ab, bb)                                                                                   \
  do {                                                                                                  \
    const int sB1 = ring_slot(rb + 2), sB2 = ring_slot(rb + 3), sA = ring_slot(rb + 4), sB0 = ring_slot(rb + 5); \
    PH(0, 0, ab, bb, RB(1 - bb, sB1, 0), NOP_, RB(1 - bb, sB1, 1), NOP_, RB(1 - bb, sB1, 2), NOP_,       \
       RB(1 - bb, sB1, 3), NOP_, IS(9), NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_);                      \
    PH(0, 1, ab, 1 - bb, RB(bb, sB2, 0), NOP_, RB(bb, sB2, 1), NOP_, RB(bb, sB2, 2), NOP_,               \
       RB(bb, sB2, 3), NOP_, IS(10), NOP_, NOP_, IS(11), NOP_, NOP_, NOP_, NOP_);                       \
    LANDED(NH * t + 11 < htot, 8);                                                                      \
    PH(0, 2, ab, bb, RA(1 - ab, sA, 0), RA(1 - ab, sA, 1), RA(1 - ab, sA, 2), RA(1 - ab, sA, 3),        \
       RA(1 - ab, sA, 4), RA(1 - ab, sA, 5), RA(1 - ab, sA, 6), RA(1 - ab, sA, 7),                      \
       RB(1 - bb, sB0, 0), RB(1 - bb, sB0, 1), RB(1 - bb, sB0, 2), RB(1 - bb, sB0, 3), IS(12), NOP_, NOP_, NOP_); \
  } while (0)
    while (true) {
      if constexpr (V == 1) STEP1(0); else if constexpr (V == 2) STEP2(0); else STEP3(0, 0);
      NEXT_T();
      if (t >= nk) break;
      if constexpr (V == 1) STEP1(1); else if constexpr (V == 2) STEP2(1); else STEP3(1, 1);
      NEXT_T();
      if (t >= nk) break;
    }
    BARRIER();
#undef STEP1
#undef STEP2
#undef STEP3
#undef NEXT_T
#undef MF
#undef RA
#undef RB
#undef IS
#undef PH
#undef NOP_
  } else {
  // Stagger: the second wave of every SIMD (waves 4-7 = wm 1) runs half a phase behind the first, so
  // one group's LDS reads overlap the other group's MFMAs. A phase is {DMA issue, fragment reads,
  // BARRIER, 16 MFMAs, BARRIER}; the late group executes one extra barrier here and the early group
  // one after the loop. Consequences for the hand-offs (both checked against the half-phase skew):
  //  - a vmcnt wait sits BEFORE the first barrier of the last phase and the data is first read in the
  //    next phase 1: two barriers later for the early group, so the late group's wait has happened;
  //  - BARRIER() retires the wave's own LDS reads (lgkmcnt(0)) first, so a slot read in phase P by the
  //    late group is safe to re-fill by the early group's DMA in phase P+1.
  if (wm == 1) BARRIER();

  for (int t = 0; t < nk; ++t) {
    if constexpr (V == 1) {         // 128x256: A B0 | B1
      ISSUE(9);                     // slot of B1(t-1), read in the previous phase 2
      READ_B(0, ring_slot(rb + 1));
      READ_A(0, ring_slot(rb));
      MFMA_PART(0, 0, 0);
      ISSUE(10); ISSUE(11);         // slots of A(t), B0(t)
      LANDED(NH * t + 11 < htot, 12);
      READ_B(1, ring_slot(rb + 2));
      MFMA_PART(0, 1, 1);
    } else if constexpr (V == 2) {  // 256x256: A0 B0 | B1 | A1 | -
      READ_B(0, ring_slot(rb + 1));
      READ_A(0, ring_slot(rb));
      MFMA_PART(0, 0, 0);
      ISSUE(10); ISSUE(11);         // slots of A0(t), B0(t)
      READ_B(1, ring_slot(rb + 2));
      MFMA_PART(0, 1, 1);
      ISSUE(12);                    // slot of B1(t)
      READ_A(0, ring_slot(rb + 3));
      MFMA_PART(1, 1, 1);
      ISSUE(13);                    // slot of A1(t)
      LANDED(NH * t + 13 < htot, 12);
      MFMA_PART(1, 0, 0);
    } else {                        // 128x384: A B0 | B1 | B2
      ISSUE(9);                     // slot of B2(t-1)
      READ_B(0, ring_slot(rb + 1));
      READ_A(0, ring_slot(rb));
      MFMA_PART(0, 0, 0);
      ISSUE(10); ISSUE(11);         // slots of A(t), B0(t)
      READ_B(1, ring_slot(rb + 2));
      MFMA_PART(0, 1, 1);
      ISSUE(12);                    // slot of B1(t)
      LANDED(NH * t + 12 < htot, 10);
      READ_B(0, ring_slot(rb + 3));
      MFMA_PART(0, 2, 0);
    }
    rb += NH;
    rb = rb >= RING ? rb - RING : rb;
  }
  if (wm == 0) BARRIER();
  }
#undef STAGE_I
#undef ISSUE
#undef STAGE_A
#undef STAGE_B
#undef READ_A
#undef READ_B
#undef MFMA_Q
#undef MFMA_PART
#undef PHASE
#undef LANDED

#if NT_DBG & 16
  if (blockIdx.x == 17 && tid == 0) {
    const unsigned long long dc = __builtin_readcyclecounter() - dbg_c0, dr = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    printf("K loop: %llu cycles in %llu ticks of 100 MHz -> %.0f MHz, %.3f us per K-tile\n", dc, dr,
           (double)dc / ((double)dr / 100.0), (double)dr / 100.0 / nk);
  }
#endif
  if constexpr (FP8) {
    // The fp8 K loop is the kernel's register peak (operands travel as 8-register tuples): the lane coordinates the epilogue
    // needs are re-derived here from mbcnt behind an opaque copy instead of being carried (and spilled) across the loop.
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    tid = lane + 64 * uw;
    frow = lane & 15; fq = lane >> 4;
  }
  // ---- epilogue. Loads are batched per 16-row slab — all bias vectors once, then the residual / aux
  // segments of one slab together — so a slab costs ONE memory round trip instead of one per 16x16
  // tile. bf16 outputs leave through LDS: in MFMA layout one store instruction is 16 rows x 32 B, i.e. 16
  // partial-line writes, and the write path of a CU retires those at a few clocks each (measured: 7 us
  // of a 28 us 16384x768x768 launch, 19 of 79 us at N = 2304). The tile is packed into a padded LDS
  // image (the ring is free now) and written out as full 128-B lines, 16 B per lane.
  // Big-tile shapes have N % TN == 0; rows >= Mstore are computed but not stored.
#if NT_DBG & 32
  st_[2] = __builtin_amdgcn_s_memrealtime();
#endif
  // image row stride (elements): TN + 8, i.e. a row stride of 4 dwords modulo the 64 banks — the 16 rows x 4 lanes of a
  // half-wave's 8-byte accesses in MFMA layout (ds_write_b64 / ds_read_b64: rows frow, 2 dwords per lane) then cover
  // the 64 banks exactly once. With TN + 16 (8 dwords modulo 64) rows r and r + 8 met in the same banks: SQ_LDS_BANK_CONFLICT
  // 1.6e8 cycles per step over the NT kernels, none in the token-major kernel that has no such image (profiles/r03_sq_counters.csv).
#ifndef NT_OROW_PAD
#define NT_OROW_PAD 8
#endif
  constexpr int OROW = TN + NT_OROW_PAD;
  static_assert(TM * OROW <= RING * HT, "output image must fit the ring");
  f32x4 csum[NBH][2];  // column sums of this wave's rows (only when p.colpart is set)
  float4 bz[NBH][2];
#pragma unroll
  for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      csum[nh][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      bz[nh][ni] = p.bias ? *(const float4*)(p.bias + bn * TN + nh * 128 + wn * 32 + ni * 16 + fq * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  const int ncol0 = bn * TN + wn * 32 + fq * 4;
  float deq = 1.0f;  // FP8: product of the operands' per-tensor dequantisation factors (1 / scale), device resident
  if constexpr (FP8) deq = p.deq_a[0] * p.deq_b[0];
  if constexpr (ACT == 3) {
    // Fused GEMM + cross-entropy, pass 1: nothing is stored but, per row, the maximum and the sum of
    // exp(logit - maximum) over this tile's real classes, and the logit of the row's target class when it lies in
    // this tile. A row's 256 columns sit in 4 lanes (fq) of 4 waves (wn): shuffles, then a [row][wn] LDS table.
    float* const stat = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mh = 0; mh < NAH; ++mh)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int lrow = mh * 128 + wm * 64 + mi * 16 + frow;
        const int m = bm * TM + lrow;
        const int tg = (int)p.ce_tgt[m];
        float mx = -INFINITY;
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            f32x4& v = acc[mh][mi][nh][ni];
            const int n0 = ncol0 + nh * 128 + ni * 16;
            v[0] += bz[nh][ni].x; v[1] += bz[nh][ni].y; v[2] += bz[nh][ni].z; v[3] += bz[nh][ni].w;
            if ((unsigned)(tg - n0) < 4u) p.ce_tlogit[m] = v[tg - n0];
            if (n0 + 3 >= p.ce_cols) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (n0 + r < p.ce_cols) ? v[r] : -INFINITY;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
          }
        float sm = 0.f;
        if (mx > -INFINITY) {
#pragma unroll
          for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const f32x4 v = acc[mh][mi][nh][ni];
              sm += (__expf(v[0] - mx) + __expf(v[1] - mx)) + (__expf(v[2] - mx) + __expf(v[3] - mx));
            }
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {  // merge the 4 lanes (fq) that hold the same row
          const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(sm, o, 64);
          const float mm = fmaxf(mx, m2);
          sm = mm > -INFINITY ? sm * __expf(mx - mm) + s2 * __expf(m2 - mm) : 0.f;
          mx = mm;
        }
        if (fq == 0) { stat[(lrow * 4 + wn) * 2] = mx; stat[(lrow * 4 + wn) * 2 + 1] = sm; }
      }
    __syncthreads();
    const int ntile = p.N / TN;
    for (int r = tid; r < TM; r += 512) {
      float mx = -INFINITY, sm = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const float m2 = stat[(r * 4 + w4) * 2], s2 = stat[(r * 4 + w4) * 2 + 1];
        const float mm = fmaxf(mx, m2);
        sm = mm > -INFINITY ? sm * __expf(mx - mm) + s2 * __expf(m2 - mm) : 0.f;
        mx = mm;
      }
      p.ce_pmax[(size_t)(bm * TM + r) * ntile + bn] = mx;
      p.ce_psum[(size_t)(bm * TM + r) * ntile + bn] = sm;
    }
    return;
  }
  if constexpr (ACT == 5 || ACT == 6) {
    // ---- LayerNorm in the epilogue. A row of the normalised matrix spans the nbn column tiles of its row block: each
    // tile forms its rows' partial statistics over its TN columns, publishes them to its partners as tagged 8-byte
    // granules (guide: "the data IS the flag"), collects its partners' and finishes its own columns. A granule is written
    // by its producer and cleared by its one consumer, so the exchange buffer is all zero again when the launch ends (no
    // per-launch memset, no epoch argument: the launch can be captured in a graph). The wait is bounded: a time-out
    // raises *ln_err instead of hanging the device.
    static_assert(NAH == 1, "LayerNorm epilogues: 128-row tiles");
    static_assert(!FP8 || NT_LN_STAGE, "the fp8 LayerNorm forms exist in the staged form only");
    if constexpr (FP8) {   // fp8 operands: the per-tensor dequantisation first, everything below is the bf16 form's arithmetic
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[0][mi][nh][ni] *= deq;
    }
    // fp8 mode: a 1-byte image of the launch's LayerNorm-side output — form 5: e4m3 of C2 = LayerNorm(..), form 6: e5m2
    // of C = the gradient of the LayerNorm's input — for the fp8 GEMMs that consume it (next projection, weight gradient)
    // (converted in the final store loop, from the bf16 image rows as they leave: inside the second pass the conversion's
    // temporaries made the backward form spill)
    constexpr int IMG = TM * OROW * 2;                          // bytes of the output image
    float* const tab = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + IMG);   // [128 rows][4 wn][2]
    float* const tab2 = tab + 128 * 4 * 2;                                                // [128 rows][2] merged
    // LEAN (the fp8 backward form): the rows' forward statistics and merged partials are not carried in registers across
    // the passes (16 per lane) but re-read from LDS tables where they are used: with the fp8 build's register allocation
    // the 128x384 backward epilogue otherwise spilled six registers
    constexpr bool LEAN = FP8 && ACT == 6;
    typedef __attribute__((address_space(1))) unsigned long long gu64_t;
    gu64_t* const xq = (gu64_t*)(p.ln_xchg);
    const float invN = 1.0f / (float)p.N;
    // (Round 3, measured and dropped: the forward form has ~60 registers to spare, so its 24 residual loads were requested
    // from inside the last K-tile. As one burst behind the last counted wait: first pass 7.8 -> 3.9 us, K loop + 4.8 us
    // — in MFMA layout a load instruction is 16 partial lines and the burst held the waves at the next barrier. One
    // load per MFMA shadow behind per-slot branches: + 0.3 us on EVERY K-tile. A second copy of the K-tile body for the
    // last K-tile (one branch per K-tile): hipcc spilled 413 registers. tools/nt_stamps.py prints these phases.)
    // per-tile row partials (a, b): forward a = sum x, b = sum x^2 ; backward a = sum dy*gamma, b = sum dy*gamma*xhat
    // every residual / pre segment of the tile is requested up front: in MFMA layout a load instruction touches 16 rows x
    // 32 B, and one dependent round trip per 16-row slab (the plain epilogue's form) cost this epilogue ~10 us per launch
    float rmean[4], rrstd[4];
    uint2 rr[4][NBH][2], ux[4][NBH][2];
    // (Also measured, not kept: the forward form's residual tile appended to the K loop's own half-tile stream — 6 slabs in
    // the half-tile image geometry, issued in the ISSUE slots that have run out of A / B half-tiles, counted by the same
    // vmcnt waits, read out of the ring into registers by conflict-free 8-byte reads. First pass 7.6 -> 4.1 us, K loop
    // + 1.6 ... 2.2 us; in the step 1.160 -> 1.150 ms for the class, 9.649 vs 9.653 ms per step: nothing.)
    // (Also measured, not kept: touching the backward form's pre-LayerNorm tile from inside the K loop — one 4-byte load
    // per 128-byte line, 4 or 8 K-tiles before the end — so that this DMA would find it in L2: the class got slower,
    // 1.583 -> 1.605 / 1.65 ms per step; loads return in order, so the touches sit in front of the K loop's counted waits.)
    // ---- staged operands (NT_LN_STAGE): one DMA instruction per tile row (TN / 8 active lanes x 16 B = the row's TN
    // bf16), wave w taking rows w, w + 8, ...; destination = the row's place in the output image (row stride OROW), so
    // every lane later finds its 8-byte segments where it will write its results: both passes work IN PLACE.
    // R2 (behind the tables): 64 rows, for the residual of form 6 in two halves — rows of mi 0,1 then rows of mi 2,3 —
    // the second half travelling under the first half's arithmetic.
    constexpr int R2OFF = IMG + 128 * 4 * 2 * 4 + 128 * 2 * 4;   // bytes
    constexpr int GAMOFF = R2OFF + 64 * OROW * 2;                 // this tile's TN gamma values (form 6), fp32
    constexpr int TAB3OFF = GAMOFF + TN * 4;                      // LEAN: [128 rows][2] = the rows' forward mean | rstd
    static_assert(GAMOFF + (ACT == 6 ? TN * 4 : 0) + (LEAN ? 1024 : 0) <= RING * HT * 2, "staging regions must fit the LDS");
    float* const tab3 = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + TAB3OFF);
    if constexpr (LEAN) {
      if (tid < TM) { tab3[tid * 2] = p.ln_mean[bm * TM + tid]; tab3[tid * 2 + 1] = p.ln_rstd[bm * TM + tid]; }
    }
    // form 6 reads gamma per 4-column group inside both passes, and those passes are pinned group by group (registers):
    // from global memory every group waited out an L2 round trip; the tile's gamma row is fetched once, here, beside the
    // operand tiles' DMA
    float* const gam_lds = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + GAMOFF);
    if constexpr (ACT == 6 && NT_LN_STAGE) {
      if (tid < TN / 4) reinterpret_cast<float4*>(gam_lds)[tid] = *(const float4*)(p.ln_gamma + bn * TN + tid * 4);
    }
    bf16_t* const r2 = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(smem) + R2OFF);
    const uint32_t stg_voff = (uint32_t)lane * 16u;
    auto stage_rows = [&](const bf16_t* src, int ld, int nrows, uint32_t lds_base, int half) {
      // half < 0: tile rows 0 .. nrows-1 in order; half 0 / 1: the 64 rows {wm' * 64 + half * 32 + q}, stored as row wm' * 32 + q
      for (int i = 0; i < nrows / 8; ++i) {
        const int q = i * 8 + uw;
        const int trow = half < 0 ? q : (q >> 5) * 64 + half * 32 + (q & 31);
        const char* b = reinterpret_cast<const char*>(src + (size_t)(bm * TM + trow) * ld + bn * TN);
        if (lane < TN / 8) DMA16(b, stg_voff, lds_base + (uint32_t)q * (OROW * 2));
      }
    };
    const uint32_t img_lds = LDS_ADDR(&smem[0]), r2_lds = img_lds + R2OFF;
    if constexpr (NT_LN_STAGE) {
      if constexpr (ACT == 5) {
        if (p.res) { stage_rows(p.res, p.ldr, TM, img_lds, -1); DMA_WAIT(); }
        __syncthreads();
      } else {
        stage_rows(p.aux, p.ldaux, TM, img_lds, -1);
        if (p.res) stage_rows(p.res, p.ldr, 64, r2_lds, 0);
        DMA_WAIT();
        __syncthreads();
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = bm * TM + wm * 64 + mi * 16 + frow;
      if constexpr (!NT_LN_STAGE) {
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr (ACT == 5) { if (p.res) rr[mi][nh][ni] = *(const uint2*)(p.res + (size_t)m * p.ldr + ncol0 + nh * 128 + ni * 16); }
            if constexpr (ACT == 6) ux[mi][nh][ni] = *(const uint2*)(p.aux + (size_t)m * p.ldaux + ncol0 + nh * 128 + ni * 16);
          }
      }
      if constexpr (ACT == 6 && !LEAN) { rmean[mi] = p.ln_mean[m]; rrstd[mi] = p.ln_rstd[m]; }
      if constexpr (LEAN) {   // (behind the staging barrier above: tab3 is complete)
        const float2 ms = *(const float2*)&tab3[(wm * 64 + mi * 16 + frow) * 2];
        rmean[mi] = ms.x; rrstd[mi] = ms.y;
      }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // form 6, staged: the first pass runs per half of the rows (mi 0,1 | mi 2,3) around the residual's two halves
#pragma unroll
    for (int hf = 0; hf < ((ACT == 6 && NT_LN_STAGE) ? 2 : 1); ++hf) {
    if constexpr (ACT == 6 && NT_LN_STAGE) {
      if (p.res) {
        if (hf == 1) { DMA_WAIT(); __syncthreads(); }   // second half has landed (issued below, under the first half's sums)
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
          for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const uint2 r = *(const uint2*)&r2[(wm * 32 + mq * 16 + frow) * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4];
              f32x4& a = acc[0][hf * 2 + mq][nh][ni];
              a[0] += bf_lo(r.x); a[1] += bf_hi(r.x); a[2] += bf_lo(r.y); a[3] += bf_hi(r.y);
            }
        if (hf == 0) { __syncthreads(); stage_rows(p.res, p.ldr, 64, r2_lds, 1); }   // R2 is free: fetch the other half
      }
    }
#pragma unroll
    for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if constexpr (ACT == 6 && NT_LN_STAGE) PIN();   // one column group at a time: hoisting every group's LDS reads to the top spilled
        const float4 gz = (ACT == 6 && NT_LN_STAGE) ? *(const float4*)(gam_lds + (ncol0 - bn * TN) + nh * 128 + ni * 16)
                                                    : *(const float4*)(p.ln_gamma + ncol0 + nh * 128 + ni * 16);
#pragma unroll
        for (int mi = ((ACT == 6 && NT_LN_STAGE) ? hf * 2 : 0); mi < ((ACT == 6 && NT_LN_STAGE) ? hf * 2 + 2 : 4); ++mi) {
          const int lrow = wm * 64 + mi * 16 + frow;
          if constexpr (NT_LN_STAGE) {
            if constexpr (ACT == 6) PIN();
            const uint2 t = *(const uint2*)&smem[lrow * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4];
            if constexpr (ACT == 5) rr[mi][nh][ni] = t; else ux[mi][nh][ni] = t;
          }
          f32x4 v = acc[0][mi][nh][ni];
          if constexpr (ACT == 5) { v[0] += bz[nh][ni].x; v[1] += bz[nh][ni].y; v[2] += bz[nh][ni].z; v[3] += bz[nh][ni].w; }
          if constexpr (ACT == 5) {
            if (p.res) {
              v[0] += bf_lo(rr[mi][nh][ni].x); v[1] += bf_hi(rr[mi][nh][ni].x);
              v[2] += bf_lo(rr[mi][nh][ni].y); v[3] += bf_hi(rr[mi][nh][ni].y);
            }
          }
          // the value as a bf16 store would leave it: what the separate LayerNorm kernels read
          uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
          v = f32x4{bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};
          if constexpr (ACT == 5) {
            *(uint2*)&smem[lrow * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4] = o;   // image 1: pre
            s1[mi] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[mi] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          } else {
            const uint2 u = ux[mi][nh][ni];
            const f32x4 xh = {(bf_lo(u.x) - rmean[mi]) * rrstd[mi], (bf_hi(u.x) - rmean[mi]) * rrstd[mi],
                              (bf_lo(u.y) - rmean[mi]) * rrstd[mi], (bf_hi(u.y) - rmean[mi]) * rrstd[mi]};
            const f32x4 g = {v[0] * gz.x, v[1] * gz.y, v[2] * gz.z, v[3] * gz.w};
            s1[mi] += (g[0] + g[1]) + (g[2] + g[3]);
            s2[mi] += (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]);
          }
          acc[0][mi][nh][ni] = v;   // forward: x; backward: dy (the second pass forms dy*gamma again)
        }
      }
    }  // hf
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int lrow = wm * 64 + mi * 16 + frow;
      float a1 = s1[mi], a2 = s2[mi];
      a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
      a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
      if (fq == 0) { tab[(lrow * 4 + wn) * 2] = a1; tab[(lrow * 4 + wn) * 2 + 1] = a2; }
    }
    __syncthreads();
#if NT_DBG & 32
    unsigned long long ls_[6];
    ls_[0] = __builtin_amdgcn_s_memrealtime();
#endif
    float pa = 0.f, pb = 0.f;   // this tile's partial of row tid (threads 0..127)
    if (tid < TM) {
      pa = (tab[(tid * 4 + 0) * 2] + tab[(tid * 4 + 1) * 2]) + (tab[(tid * 4 + 2) * 2] + tab[(tid * 4 + 3) * 2]);
      pb = (tab[(tid * 4 + 0) * 2 + 1] + tab[(tid * 4 + 1) * 2 + 1]) + (tab[(tid * 4 + 2) * 2 + 1] + tab[(tid * 4 + 3) * 2 + 1]);
      if constexpr (ACT == 5) {   // (mean, M2) of the tile's TN values: merged below without cancellation
        const float mt = pa * (1.0f / (float)TN);
        pb = pb - pa * mt;
        pa = mt;
      }
      // publish: the data IS the flag — one naturally aligned 8-byte granule {value, tag = 1} per number and consumer,
      // written by ONE agent-scope store each (no flag word, no fence, no drain: a granule is either old or whole)
      const unsigned long long wa = (1ull << 32) | __float_as_uint(pa), wb = (1ull << 32) | __float_as_uint(pb);
      const bool mute = p.ln_fault == 1 && bn == nbn - 1;   // fault injection 1 (tests only): this tile never publishes
      for (int c = 0; c < nbn; ++c)
        if (c != bn && !mute) {
          gu64_t* g = xq + (((size_t)(bm * nbn + bn) * nbn + c) * TM + tid) * 2;
          __hip_atomic_store(g, wa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(g + 1, wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (ACT == 5) {   // image 1 (pre) leaves while the partners' partials arrive
      constexpr int CPR = TN / 8;
      bf16_t* const cbase = p.C + (size_t)(bm * TM) * p.ldc + bn * TN;
      const int rows_ok = p.Mstore - bm * TM;
#pragma unroll 4
      for (int c = tid; c < TM * CPR; c += 512) {
        const int r = c / CPR, cc = c - r * CPR;
        const uint4 v = *(const uint4*)&smem[r * OROW + cc * 8];
        if (r < rows_ok) {
          if constexpr (NT_OUT_NT & 8) __builtin_nontemporal_store(u32x4nt{v.x, v.y, v.z, v.w}, (u32x4nt*)(cbase + (size_t)r * p.ldc + cc * 8));
          else OUT_STORE((cbase + (size_t)r * p.ldc + cc * 8), v);
        }
      }
    }
#if NT_DBG & 32
    ls_[1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < TM) {   // collect and merge the nbn partials of row tid in tile order (the same arithmetic on every tile)
      float qa[4], qb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qa[j] = 0.f; qb[j] = 0.f;
        if (j < nbn) {
          if (j == bn) { qa[j] = pa; qb[j] = pb; }
          else {
            // consume: re-read this row's two granules of producer j (agent-scope loads: past the L1) until both carry
            // the tag, then clear them — every granule has one writer and one clearer, so the buffer is all zero
            // again when the launch ends (no memset, no epoch argument; the wait is bounded)
            gu64_t* g = xq + (((size_t)(bm * nbn + j) * nbn + bn) * TM + tid) * 2;
            unsigned long long wa = 0, wb = 0;
            bool seen = false;
            for (unsigned spins = 0; spins < (1u << 18); ++spins) {   // ~0.5 s of polling: far beyond any launch
              wa = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              wb = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((wa >> 32) == 1ull && (wb >> 32) == 1ull) { seen = true; break; }
              __builtin_amdgcn_s_sleep(4);
            }
            // A time-out makes the launch end instead of hanging the device; the error word then (1) turns this step's
            // loss into NaN and reaches the host (plb_launch_step_status), (2) makes the AdamW launch skip, and (3) stays
            // set until plb_status has reported it and re-zeroed this buffer: a producer whose store lands after its
            // consumer gave up leaves a tagged granule behind, which only that memset removes.
            const bool late = p.ln_fault == 2 && bn == 0;   // fault injection 2 (tests only): as if the stores landed after a time-out
            if (!seen || late) atomicAdd(p.ln_err, 1u);
            if (!late) {
              __hip_atomic_store(g, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            qa[j] = __uint_as_float((unsigned int)wa); qb[j] = __uint_as_float((unsigned int)wb);
          }
        }
      }
      if constexpr (ACT == 5) {
        float mean = 0.f;
        for (int j = 0; j < nbn; ++j) mean += qa[j];
        mean *= 1.0f / (float)nbn;
        float m2 = 0.f;
        for (int j = 0; j < nbn; ++j) m2 += qb[j] + (float)TN * (qa[j] - mean) * (qa[j] - mean);
        const float rstd = rsqrtf(m2 * invN + p.ln_eps);
        tab2[tid * 2] = mean; tab2[tid * 2 + 1] = rstd;
        const int m = bm * TM + tid;
        if (bn == 0 && m < p.Mstore) { p.ln_mean[m] = mean; p.ln_rstd[m] = rstd; }
      } else {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < nbn; ++j) { a += qa[j]; b += qb[j]; }
        tab2[tid * 2] = a * invN; tab2[tid * 2 + 1] = b * invN;
      }
    }
    __syncthreads();   // also: every thread's reads of image 1 have retired (its stores consumed them)
#if NT_DBG & 32
    ls_[2] = __builtin_amdgcn_s_memrealtime();
#endif
    // second pass, one 4-column group of the lane at a time (its 4 rows innermost): the three column sums of the backward
    // then need 12 registers instead of 72, which is what lets the pre segments of the first pass stay in registers
    float t0[4], t1[4];
    if constexpr (!LEAN) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { t0[mi] = tab2[(wm * 64 + mi * 16 + frow) * 2]; t1[mi] = tab2[(wm * 64 + mi * 16 + frow) * 2 + 1]; }
    }
    if constexpr (LEAN) asm volatile("" ::: "memory");   // the first pass's copies of the statistics end here
    // the lane's column offset behind an opaque copy: the second pass forms its addresses afresh from this ONE register
    // instead of carrying the first pass's 64-bit pointers across the hand-off
    int ncol2 = ncol0;
    asm volatile("" : "+v"(ncol2));
#pragma unroll
    for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if constexpr (ACT == 6 && NT_LN_STAGE) PIN();
        float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gz = (ACT == 6 && NT_LN_STAGE) ? *(const float4*)(gam_lds + (ncol2 - bn * TN) + nh * 128 + ni * 16)
                                                    : *(const float4*)(p.ln_gamma + ncol2 + nh * 128 + ni * 16);
        if constexpr (ACT == 5) bt = *(const float4*)(p.ln_beta + ncol2 + nh * 128 + ni * 16);
        f32x4 cs = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int lrow = wm * 64 + mi * 16 + frow;
          const f32x4 x = acc[0][mi][nh][ni];
          f32x4 y;
          if constexpr (LEAN) {
            const float2 tt = *(const float2*)&tab2[lrow * 2], ms = *(const float2*)&tab3[lrow * 2];
            t0[mi] = tt.x; t1[mi] = tt.y; rmean[mi] = ms.x; rrstd[mi] = ms.y;
            asm volatile("" : "+v"(t0[mi]), "+v"(t1[mi]), "+v"(rmean[mi]), "+v"(rrstd[mi]));   // not hoisted out of the loops
          }
          if constexpr (ACT == 5) {
            y = f32x4{(x[0] - t0[mi]) * t1[mi] * gz.x + bt.x, (x[1] - t0[mi]) * t1[mi] * gz.y + bt.y,
                      (x[2] - t0[mi]) * t1[mi] * gz.z + bt.z, (x[3] - t0[mi]) * t1[mi] * gz.w + bt.w};
          } else {
            const float mu = rmean[mi], rs = rrstd[mi];
            uint2 u;
            if constexpr (NT_LN_STAGE) u = *(const uint2*)&smem[lrow * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4];   // still there: dx replaces it below
            else u = ux[mi][nh][ni];
            // opaque copy: otherwise hipcc keeps the first pass's 96 unpacked xhat values alive across the hand-off
            // (common sub-expressions of the two passes) instead of the 48 packed registers, and spills
            asm volatile("" : "+v"(u.x), "+v"(u.y));
            const f32x4 xh = {(bf_lo(u.x) - mu) * rs, (bf_hi(u.x) - mu) * rs, (bf_lo(u.y) - mu) * rs, (bf_hi(u.y) - mu) * rs};
            dg += x * xh; db += x;
            y = f32x4{rs * (x[0] * gz.x - t0[mi] - xh[0] * t1[mi]), rs * (x[1] * gz.y - t0[mi] - xh[1] * t1[mi]),
                      rs * (x[2] * gz.z - t0[mi] - xh[2] * t1[mi]), rs * (x[3] * gz.w - t0[mi] - xh[3] * t1[mi])};
          }
          uint2 o; o.x = pack_bf2(y[0], y[1]); o.y = pack_bf2(y[2], y[3]);
          *(uint2*)&smem[lrow * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4] = o;
          if (ACT == 6 && bm * TM + lrow < p.Mstore) cs += f32x4{bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};

        }
        if constexpr (ACT == 6 && !(NT_DBG & 64)) {   // (NT_DBG 64: timing build without these outputs)
          // dgamma | dbeta | column sums of dx over this wave's 64 rows: one partial row per (row tile, wm)
#pragma unroll
          for (int q3 = 0; q3 < 3; ++q3) {
            f32x4 v = q3 == 0 ? dg : q3 == 1 ? db : cs;
            v = f32x4{row16_sum(v[0]), row16_sum(v[1]), row16_sum(v[2]), row16_sum(v[3])};
            if (frow == 0)
              *(float4*)(p.colpart + ((size_t)(bm * 2 + wm) * 3 + q3) * p.N + ncol2 + nh * 128 + ni * 16) =
                  make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    __syncthreads();
#if NT_DBG & 32
    ls_[3] = __builtin_amdgcn_s_memrealtime();
#endif
    {
      constexpr int CPR = TN / 8;
      bf16_t* const obase = (ACT == 5 ? p.C2 + (size_t)(bm * TM) * p.ldc2 : p.C + (size_t)(bm * TM) * p.ldc) + bn * TN;
      const int ldo = ACT == 5 ? p.ldc2 : p.ldc;
      const int rows_ok = p.Mstore - bm * TM;
      // fp8 builds with an image requested: every 16-byte chunk of the bf16 image leaves twice — as it is, and as the 8
      // bytes of the 1-byte image (e4m3 for the forward form's LayerNorm output, e5m2 for the backward form's gradient)
      const bool img8 = FP8 && p.C8 != nullptr;
      const float qs8 = img8 ? p.q_scale[0] : 1.0f;
      constexpr bool bf8o = ACT == 6;
      float amax8 = 0.f;
      unsigned char* const qbase = img8 ? p.C8 + (size_t)(bm * TM) * p.ldc8 + bn * TN : nullptr;
#pragma unroll 4
      for (int c = tid; c < TM * CPR; c += 512) {
        const int r = c / CPR, cc = c - r * CPR;
        const uint4 v = *(const uint4*)&smem[r * OROW + cc * 8];
        if (r < rows_ok) {
          OUT_STORE((obase + (size_t)r * ldo + cc * 8), v);
          if constexpr (FP8) {
            if (img8) {
              const float f0 = bf_lo(v.x), f1 = bf_hi(v.x), f2 = bf_lo(v.y), f3 = bf_hi(v.y);
              const float f4 = bf_lo(v.z), f5 = bf_hi(v.z), f6 = bf_lo(v.w), f7 = bf_hi(v.w);
              uint2 w;
              w.x = pack_fp8x4(f0 * qs8, f1 * qs8, f2 * qs8, f3 * qs8, bf8o);
              w.y = pack_fp8x4(f4 * qs8, f5 * qs8, f6 * qs8, f7 * qs8, bf8o);
              *(uint2*)(qbase + (size_t)r * p.ldc8 + cc * 8) = w;
              amax8 = fmaxf(amax8, fmaxf(fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3))),
                                         fmaxf(fmaxf(fabsf(f4), fabsf(f5)), fmaxf(fabsf(f6), fabsf(f7)))));
            }
          }
        }
      }
      if constexpr (FP8) {
        if (img8 && p.q_amax) {
          amax8 = wave_max(amax8);
          if (lane == 0) atomic_max_abs(p.q_amax, amax8, blockIdx.x * 8 + uw);
        }
      }
    }
#if NT_DBG & 32
    ls_[4] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ls_[5] = __builtin_amdgcn_s_memrealtime();
    if ((blockIdx.x % 256) == 17 && tid == 0)
      printf("blk %d LN form %d: fill %llu | kloop %llu | pass 1 %llu | publish + image 1 out %llu | partners' partials %llu | pass 2 %llu | "
             "stores issued %llu | drained %llu (x10 ns)\n", (int)blockIdx.x, ACT, st_[1] - st_[0], st_[2] - st_[1], ls_[0] - st_[2],
             ls_[1] - ls_[0], ls_[2] - ls_[1], ls_[3] - ls_[2], ls_[4] - ls_[3], ls_[5] - ls_[4]);
#endif
    return;
  }
  // Forms 7 / 8: the gelu'(u) stash is private to this pair of launches (same tiles, same lane roles), so it lives in
  // LANE layout — tile t, slab (mh, mi), column half nh: 512 consecutive 16-byte items, one per thread = {ni 0, ni 1} of
  // the lane's 4-column groups. Form 7 stores it straight from the registers (no LDS image, no second store pass),
  // form 8 reads it back as full 1-KiB wave accesses. Row-major, a lane's 8-byte segments were 16 partial lines per load
  // instruction: 8.0 us of form 8's epilogue against 2.7 us of arithmetic (tools/nt_stamps.py).
  uint4* const lane_stash = reinterpret_cast<uint4*>(ACT == 7 ? p.C : const_cast<bf16_t*>(p.aux));
  constexpr int SQD = 3;       // stash slabs requested ahead (ring of SQD + 1)
  uint4 sq[SQD + 1][NBH];      // ACT == 8: stash items in flight
  uint2 keep[NAH][4][NBH][2];  // ACT == 1: the packed pre-activations, for the second (gelu) image
  // FP8 with p.C8: the fp8 copy of the output (act 1: of gelu) that the next fp8 GEMM reads, packed 4 values per
  // register here and written through the LDS image as bytes after the bf16 outputs have left; amax of the launch
  uint32_t k8[NAH][4][NBH][2];
  const bool want8 = FP8 && p.C8 != nullptr;
  const float qs = want8 ? p.q_scale[0] : 1.0f;
  float amax = 0.f;
  // fp8 mode: an output that only fp8 GEMMs consume (gelu(u), dU) leaves as its 1-byte image alone — no bf16 image
  const bool want16 = !FP8 || (ACT == 7 ? p.C2 : p.C) != nullptr;
#pragma unroll
  for (int mh = 0; mh < NAH; ++mh)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int lrow = mh * 128 + wm * 64 + mi * 16 + frow;  // row inside the tile
      const int m = bm * TM + lrow;
      uint2 rr[NBH][2], ux[NBH][2];
      if (p.res) {
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            rr[nh][ni] = *(const uint2*)(p.res + (size_t)m * p.ldr + ncol0 + nh * 128 + ni * 16);
      }
      if (ACT == 2) {
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            ux[nh][ni] = *(const uint2*)(p.aux + (size_t)m * p.ldaux + ncol0 + nh * 128 + ni * 16);
      }
      if (ACT == 8) {
        // the stash of form 7 in LANE layout (see there): this lane's 16 bytes of slab (mh, mi), column half nh — requested
        // SQD slabs ahead: taken slab by slab, each of the 8 slabs of a tile cost one memory round trip
        constexpr int NS = NAH * 4;
        const int sl = mh * 4 + mi;
        if (sl == 0) {
#pragma unroll
          for (int a = 0; a < SQD && a < NS; ++a)
#pragma unroll
            for (int nh = 0; nh < NBH; ++nh)
              sq[a][nh] = lane_stash[((size_t)(bm * nbn + bn) * (NS * NBH) + a * NBH + nh) * 512 + tid];
        }
        if (sl + SQD < NS) {
#pragma unroll
          for (int nh = 0; nh < NBH; ++nh)
            sq[(sl + SQD) % (SQD + 1)][nh] = lane_stash[((size_t)(bm * nbn + bn) * (NS * NBH) + (sl + SQD) * NBH + nh) * 512 + tid];
        }
#pragma unroll
        for (int nh = 0; nh < NBH; ++nh) {
          const uint4 q = sq[sl % (SQD + 1)][nh];
          ux[nh][0] = make_uint2(q.x, q.y); ux[nh][1] = make_uint2(q.z, q.w);
        }
      }
      const bool st = m < p.Mstore;
      // ACT == 4, fused GEMM + cross-entropy pass 2: the logits are recomputed and leave as the gradient
      // (softmax - onehot) * w, with the row's log-sum-exp from pass 1; padding columns and rows get 0 (w = 0)
      float ce_l = 0.f, ce_wt = 0.f;
      int ce_t = -1;
      if (ACT == 4) { ce_l = p.ce_lse[m]; ce_wt = p.ce_w[m]; ce_t = (int)p.ce_tgt[m]; }
      uint2 dpair = make_uint2(0u, 0u);
#pragma unroll
      for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          f32x4 v = acc[mh][mi][nh][ni];
          const int n0 = ncol0 + nh * 128 + ni * 16;
          if constexpr (FP8) { v[0] *= deq; v[1] *= deq; v[2] *= deq; v[3] *= deq; }
          v[0] += bz[nh][ni].x; v[1] += bz[nh][ni].y; v[2] += bz[nh][ni].z; v[3] += bz[nh][ni].w;
          if (ACT == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // select, never 0 * exp(): rows without loss (w = 0: padding, positions past the length) hold live or
              // stale hidden states whose logit could overflow the exponential
              float gq = (ce_wt != 0.f && n0 + r < p.ce_cols) ? __expf(v[r] - ce_l) * ce_wt : 0.f;
              gq -= (n0 + r == ce_t) ? ce_wt : 0.f;
              v[r] = gq;
            }
          }
          if (p.res) {
            v[0] += bf_lo(rr[nh][ni].x); v[1] += bf_hi(rr[nh][ni].x);
            v[2] += bf_lo(rr[nh][ni].y); v[3] += bf_hi(rr[nh][ni].y);
          }
          if (ACT == 2) {  // gelu backward: multiply by gelu_new'(u); exact zeros stay zeros whatever u holds
            v[0] = v[0] != 0.f ? v[0] * gelu_new_grad_f(bf_lo(ux[nh][ni].x)) : 0.f;
            v[1] = v[1] != 0.f ? v[1] * gelu_new_grad_f(bf_hi(ux[nh][ni].x)) : 0.f;
            v[2] = v[2] != 0.f ? v[2] * gelu_new_grad_f(bf_lo(ux[nh][ni].y)) : 0.f;
            v[3] = v[3] != 0.f ? v[3] * gelu_new_grad_f(bf_hi(ux[nh][ni].y)) : 0.f;
          }
          if (ACT == 8) {  // gelu backward on a stashed DERIVATIVE (form 7 below): one multiply per value
            v[0] = v[0] != 0.f ? v[0] * bf_lo(ux[nh][ni].x) : 0.f; v[1] = v[1] != 0.f ? v[1] * bf_hi(ux[nh][ni].x) : 0.f;
            v[2] = v[2] != 0.f ? v[2] * bf_lo(ux[nh][ni].y) : 0.f; v[3] = v[3] != 0.f ? v[3] * bf_hi(ux[nh][ni].y) : 0.f;
          }
          uint2 gpk = make_uint2(0u, 0u);
          if (ACT == 7) {
            // gelu forward that stashes gelu_new'(u) instead of u: the sigmoid (one v_exp, one v_rcp) is shared by the
            // activation and its derivative, so the backward epilogue multiplies by a stashed number instead of
            // evaluating the derivative (128 evaluations per lane of a 2-waves-per-SIMD epilogue). Both are formed from
            // the fp32 pre-activation; u itself is not kept (nothing but the derivative ever read it).
            f32x4 gg;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = v[r], x2 = x * x;
              const float sg = gelu_sigmoid(x, x2);
              gg[r] = x * sg;
              v[r] = __builtin_fmaf(sg, x * __builtin_fmaf(x2, 0.21406444f, 1.5957691f) * (1.0f - sg), sg);
            }
            gpk.x = pack_bf2(gg[0], gg[1]); gpk.y = pack_bf2(gg[2], gg[3]);
            asm volatile("" : "+v"(gpk.x), "+v"(gpk.y));  // packed NOW: carried in fp32 to the second image it spilled
          }
          if (OUTF32) {
            if (st) *(float4*)(p.Cf + (size_t)m * p.ldcf + n0) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            if (ACT == 7) {   // derivative -> lane-layout stash (both ni of a column half make one 16-byte item); image <- activation
              if (ni == 0) dpair = o;
              else   // read once, a backward pass later: non-temporal like the other outputs of the gelu forms
                __builtin_nontemporal_store(u32x4nt{dpair.x, dpair.y, o.x, o.y},
                    (u32x4nt*)(lane_stash + ((size_t)(bm * nbn + bn) * (NAH * 4 * NBH) + (mh * 4 + mi) * NBH + nh) * 512 + tid));
              if (want16) *(uint2*)&smem[lrow * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4] = gpk;
            } else {
              if (want16) *(uint2*)&smem[lrow * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4] = o;
            }
            if (ACT == 1) keep[mh][mi][nh][ni] = o;
            v = f32x4{bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};  // the values as stored
            if constexpr (FP8 && ACT != 1) {
              if (want8) {   // form 7: the image is of gelu(u) (what the bf16 path's FFN output GEMM reads), not of the stash
                const f32x4 z = ACT == 7 ? f32x4{bf_lo(gpk.x), bf_hi(gpk.x), bf_lo(gpk.y), bf_hi(gpk.y)} : v;
                k8[mh][mi][nh][ni] = pack_fp8x4(z[0] * qs, z[1] * qs, z[2] * qs, z[3] * qs, ACT == 7 ? false : ACT == 8 ? true : p.c8_bf8 != 0);
                if (st) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(z[0]), fabsf(z[1])), fmaxf(fabsf(z[2]), fabsf(z[3]))));
              }
            }
          }
          if (!NOCS && st) csum[nh][ni] += v;
        }
    }
  if (!OUTF32) {
    constexpr int CPR = TN / 8;  // 16-B chunks per tile row
    const int ldimg = ACT == 7 ? p.ldc2 : p.ldc;   // form 7: the image is gelu(u) -> C2 (C is the lane-layout stash)
    bf16_t* const cbase = (ACT == 7 ? p.C2 : p.C) + (size_t)(bm * TM) * ldimg + bn * TN;
    const int rows_ok = p.Mstore - bm * TM;  // rows of this tile that are stored
    __syncthreads();
#if NT_DBG & 32
    st_[3] = __builtin_amdgcn_s_memrealtime();
#endif
    if (want16) {
#pragma unroll 4
      for (int c = tid; c < TM * CPR; c += 512) {
        const int r = c / CPR, cc = c - r * CPR;
        const uint4 v = *(const uint4*)&smem[r * OROW + cc * 8];
        if (r < rows_ok) OUT_STORE((cbase + (size_t)r * ldimg + cc * 8), v);
      }
    }
    if (ACT == 1) {  // gelu forward: C keeps the bf16 pre-activation u, C2 = gelu_new(u)
      __syncthreads();
#pragma unroll
      for (int mh = 0; mh < NAH; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const uint2 o = keep[mh][mi][nh][ni];
              uint2 g;
              g.x = pack_bf2(gelu_new_f(bf_lo(o.x)), gelu_new_f(bf_hi(o.x)));
              g.y = pack_bf2(gelu_new_f(bf_lo(o.y)), gelu_new_f(bf_hi(o.y)));
              *(uint2*)&smem[(mh * 128 + wm * 64 + mi * 16 + frow) * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4] = g;
              if constexpr (FP8) {
                if (want8) {  // the fp8 image of gelu(u) as stored in bf16: what the bf16 path's FFN2 would read
                  const float g0 = bf_lo(g.x), g1 = bf_hi(g.x), g2 = bf_lo(g.y), g3 = bf_hi(g.y);
                  k8[mh][mi][nh][ni] = pack_fp8x4(g0 * qs, g1 * qs, g2 * qs, g3 * qs, p.c8_bf8 != 0);
                  if (bm * TM + mh * 128 + wm * 64 + mi * 16 + frow < p.Mstore)
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fmaxf(fabsf(g2), fabsf(g3))));
                }
              }
            }
      __syncthreads();
      bf16_t* const gbase = p.C2 + (size_t)(bm * TM) * p.ldc2 + bn * TN;
#pragma unroll 4
      for (int c = tid; c < TM * CPR; c += 512) {
        const int r = c / CPR, cc = c - r * CPR;
        const uint4 v = *(const uint4*)&smem[r * OROW + cc * 8];
        if (r < rows_ok) OUT_STORE((gbase + (size_t)r * p.ldc2 + cc * 8), v);
      }
    }
    if constexpr (FP8) {
      if (want8) {  // third image: bytes, row stride OROW bytes; full 16-byte chunks out
        unsigned char* const img8 = reinterpret_cast<unsigned char*>(smem);
        __syncthreads();
#pragma unroll
        for (int mh = 0; mh < NAH; ++mh)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                *(uint32_t*)&img8[(mh * 128 + wm * 64 + mi * 16 + frow) * OROW + nh * 128 + wn * 32 + ni * 16 + fq * 4] =
                    k8[mh][mi][nh][ni];
        __syncthreads();
        constexpr int CPR8 = TN / 16;
        unsigned char* const qbase = p.C8 + (size_t)(bm * TM) * p.ldc8 + bn * TN;
#pragma unroll 4
        for (int c = tid; c < TM * CPR8; c += 512) {
          const int r = c / CPR8, cc = c - r * CPR8;
          const uint4 v = *(const uint4*)&img8[r * OROW + cc * 16];
          if (r < rows_ok) *(uint4*)(qbase + (size_t)r * p.ldc8 + cc * 16) = v;
        }
        if (p.q_amax) {
          amax = wave_max(amax);
          if (lane == 0) atomic_max_abs(p.q_amax, amax, blockIdx.x * 8 + uw);
        }
      }
    }
  }
  if (!NOCS && p.colpart) {
    // Bias gradient of the producing Linear for free: sum the stored values over this wave's 64*NAH rows
    // (16 lanes hold 16 different rows of the same 4 columns) and write ONE partial row per (row tile, wm):
    // colpart[(bm*2 + wm)][n]; a fixed-order column sum over these few rows finishes it (deterministic).
#pragma unroll
    for (int nh = 0; nh < NBH; ++nh)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f32x4 v = csum[nh][ni];
        v = f32x4{row16_sum(v[0]), row16_sum(v[1]), row16_sum(v[2]), row16_sum(v[3])};
        if (frow == 0)
          *(float4*)(p.colpart + (size_t)(bm * 2 + wm) * p.N + bn * TN + nh * 128 + wn * 32 + ni * 16 + fq * 4) =
              make_float4(v[0], v[1], v[2], v[3]);
      }
  }
#if NT_DBG & 32
  st_[4] = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st_[5] = __builtin_amdgcn_s_memrealtime();
  if ((blockIdx.x % 256) == 17 && tid == 0)
    printf("blk %d: start %llu | fill %llu | kloop %llu | image %llu | stores issued %llu | drained %llu (x10 ns)\n", (int)blockIdx.x,
           st_[0] % 1000000ull, st_[1] - st_[0], st_[2] - st_[1], st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4]);
#endif
}

}  // namespace
