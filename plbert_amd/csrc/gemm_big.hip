// Large-tile NT GEMMs with a multi-phase, counted-vmcnt LDS-DMA pipeline (gfx950).
//
// Why: at a 128x128 tile the operand stream into LDS is 64 FLOP/B, and a CU's LDS-DMA path (one 1 KiB
// global_load_lds per 16 clocks through the texture addresser: ~94 GB/s per CU measured) bounds the kernel
// long before the MFMA pipe. Here one workgroup of 8 waves (2 per SIMD) owns a CU, the tile is 256x256
// (128 FLOP/B), 128x384 (96 FLOP/B) or 128x256 (85 FLOP/B), and 4-6 16-KiB half-tiles of LDS-DMA stay in
// flight across barriers (counted s_waitcnt vmcnt(N), never 0 in steady state). Tile choice is by wave
// quantisation on 256 CUs: N = 768 -> 128x384 gives exactly 256 tiles at M = 16384 (256x256 would give 192:
// a quarter of the chip idle), N = 2304 -> 768 tiles = 3 per CU, N = 2048 -> 512 tiles of 256x256.
//
// A K-tile (64 deep) is 3-4 128-row half-tiles, streamed through a 10-slot ring (the whole 160 KiB LDS) in
// consumption order:   256x256: A0 B0 B1 A1      128x384: A B0 B1 B2      128x256: A B0 B1
// 8 waves = 2 (M) x 4 (N). Wave (wm, wn) owns rows {mh*128 + wm*64 + 0..63} and columns
// {nh*128 + wn*32 + 0..31} of every (mh, nh): its output is interleaved over the half-tiles, so one
// phase (16 MFMAs 16x16x32 = a 64x32 patch) reads fragments of ONE A half and ONE B half:
//   256x256:  ph1 A0,B0 -> (0,0) | ph2 B1 -> (0,1) | ph3 A1 -> (1,1) | ph4 (1,0)
//   128x384:  ph1 A,B0 -> nh 0   | ph2 B1 -> nh 1  | ph3 B2 -> nh 2
// A slot is re-filled once the phase that read its previous occupant is behind a barrier. Barriers are raw
// s_barrier (asm: __syncthreads() would drain vmcnt to 0). Two forms of the K loop exist, chosen per launch
// by measurement (plb_launch_gemm_nt_big): "interleaved" (fragment reads of the next phase and the DMA
// issues dealt into the MFMA shadows, one barrier per phase) and "staggered" (two barriers per phase, the
// two waves of a SIMD half a phase apart) — see the comments at the loops.
#include "gemm_nt_pipeline.h"

// Timing experiments only (tools/build_dbg.sh): -DNT_DBG=<bit mask> 1 no MFMA, 2 no fragment reads, 4 no DMA
// after the prologue, 8 no barriers in the K loop (results are garbage with any of those), 16 print the shader
// clock over the K loop (results stay valid). The shipped library has 0.
namespace {

template <int V, bool PF>
int launch_big(const PlbGemmNT* p, int act, int out_f32, hipStream_t stream) {
  constexpr int TM = (V == 2 ? 2 : 1) * 128, TN = (V == 3 ? 3 : 2) * 128;
  if (p->M % TM || p->N % TN || p->K % 64 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 1;
  dim3 grid((p->M / TM) * (p->N / TN)), block(512);
  if (out_f32) {
    if (act != 0) return 1;
    hipLaunchKernelGGL((gemm_nt_big_kernel<V, 0, true, PF>), grid, block, 0, stream, *p);
  } else if (act == 0) {
    if (p->colpart) hipLaunchKernelGGL((gemm_nt_big_kernel<V, 0, false, PF>), grid, block, 0, stream, *p);
    else hipLaunchKernelGGL((gemm_nt_big_kernel<V, 0, false, PF, false, false, true>), grid, block, 0, stream, *p);
  } else if (act == 1) {
    hipLaunchKernelGGL((gemm_nt_big_kernel<V, 1, false, PF>), grid, block, 0, stream, *p);
  } else if (act == 2) {
    hipLaunchKernelGGL((gemm_nt_big_kernel<V, 2, false, PF>), grid, block, 0, stream, *p);
  } else if ((act == 3 || act == 4) && TN == 256) {  // fused GEMM + cross-entropy passes: 256-column tiles only
    if (!p->ce_tgt || p->ce_cols < 1 || p->ce_cols > p->N) return 1;
    if (act == 3) {
      if (!p->ce_pmax || !p->ce_psum || !p->ce_tlogit) return 1;
      hipLaunchKernelGGL((gemm_nt_big_kernel<V, 3, false, PF>), grid, block, 0, stream, *p);
    } else {
      if (!p->ce_lse || !p->ce_w || !p->C) return 1;
      hipLaunchKernelGGL((gemm_nt_big_kernel<V, 4, false, PF>), grid, block, 0, stream, *p);
    }
  } else {
    return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---------------------------------------------------------------------------------------------------
// dW[N,K] = A[rows,N]^T · B[rows,K]: the token-major ("TN") weight-gradient GEMM on the same pipeline.
// Output tile 256 (n) x 256 (k); the reduction runs over token rows t in steps of 64 and is split over
// blockIdx.y into fp32 slabs. A half-tile is [64 t][128 cols] = two [64][64] LDS images with 128-B
// rows filled by LDS-DMA in full 128-B lines. MFMA fragments need the reduction index on the k
// axis, i.e. COLUMNS of these images: ds_read_b64_tr_b16. A half-wave's transposed read touches rows
// {r..r+3, r+8..r+11} x 32 B; XOR-ing the 32-B pair index with ((row>>1)&1) | ((row>>3)&1)<<1 spreads
// those eight segments over all 64 banks (the swizzle is applied to the DMA's per-lane source chunk).
DEVI int tn_g(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// 224 VGPRs: two waves of this kernel then leave 64 registers of every SIMD (and 32 KiB of LDS) to the side stream's small
// kernels, which run beside it on the same CUs (see the K loop)
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(224))) void gemm_tn_big_kernel(PlbGemmTN p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * HT];  // [buf][A0,A1,B0,B1][2 images][64][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = uw >> 2, wn = uw & 3;
  // Work placement: every tile of one row split streams the same token rows, so the tiles of a split
  // should share an L2. Blocks are dealt round-robin over the 8 XCDs (block id % 8 labels the XCD
  // group — speed only, never correctness): XCD x takes splits [x*s, (x+1)*s), all their tiles.
  const int nbk = p.K >> 8;
  const int tiles = (p.Ncols >> 8) * nbk;
  // xcd_remap hands XCD x the x-th contiguous run of the logical order (split-major): whole splits when their count
  // is a multiple of 8, otherwise a split may straddle two XCDs (its rows are then fetched by both L2s).
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int split = logical / tiles, tile = logical % tiles;
  const int bn = tile / nbk, bk = tile % nbk;
  const int t_begin = split * p.rows_per_split;
  int t_end = t_begin + p.rows_per_split;
  if (t_end > p.Mtot) t_end = p.Mtot;
  const int nk = (t_end - t_begin) >> 6;

  // ---- staging: instruction i = 2w + j of a half-tile covers image i>>3, rows (i&7)*8 + (lane>>3)
  const int i0 = 2 * uw, i1 = 2 * uw + 1;
  const int r0 = (i0 & 7) * 8 + (lane >> 3), r1 = (i1 & 7) * 8 + (lane >> 3);
  const int sc0 = (((lane & 7) ^ (2 * tn_g(r0))) * 8) + (i0 >> 3) * 64;  // source column within the half
  const int sc1 = (((lane & 7) ^ (2 * tn_g(r1))) * 8) + (i1 >> 3) * 64;
  const int dst0 = (i0 >> 3) * 4096 + (i0 & 7) * 8 * 64, dst1 = (i1 >> 3) * 4096 + (i1 & 7) * 8 * 64;
  // ---- fragment reads (transposed): lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3;
  // k-step kk covers t rows kk*32 + 8*(lane>>4) + {0..3} (first read) and +4 (second read)
  const int fg = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  // A fragment (mi): image wm of the half, columns mi*16 + 4p ; B fragment (ni): image wn>>1, columns
  // (wn&1)*32 + ni*16 + 4p. The swizzle term of row kk*32 + 8*fg + 4*s2 + q depends on the lane only
  // (((row>>1)&1) = (q>>1)&1, ((row>>3)&1) = fg&1), so every read is lane base + compile-time offset.
  const int gsw = 2 * (((q4 >> 1) & 1) | ((fg & 1) << 1));
  const int rowbase = (8 * fg + q4) * 64 + (p4 & 1) * 4;
  int aoff[4], boff[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) aoff[mi] = wm * 4096 + rowbase + (((2 * mi + (p4 >> 1)) ^ gsw) << 3);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
    boff[ni] = (wn >> 1) * 4096 + rowbase + (((4 * (wn & 1) + 2 * ni + (p4 >> 1)) ^ gsw) << 3);
  const unsigned lds0 = (unsigned)(size_t)&smem[0];
#define FRAG(r, kk) (bf16x8{r[2 * (kk)][0], r[2 * (kk)][1], r[2 * (kk)][2], r[2 * (kk)][3],              \
                            r[2 * (kk) + 1][0], r[2 * (kk) + 1][1], r[2 * (kk) + 1][2], r[2 * (kk) + 1][3]})

  f32x4 acc[2][4][2][2];  // [mh][mi][nh][ni]: n = mh*128 + wm*64 + mi*16 + .., k = nh*128 + wn*32 + ni*16 + ..
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
// TN_DBG (timing builds only, tools/tn_wait.py): 1 = print, from one workgroup, the share of the K loop a wave spends in the
// vmcnt wait that precedes phase 4 (is the loop waiting for memory?) and the loop's time per K-tile
#ifndef TN_DBG
#define TN_DBG 0
#endif
#if TN_DBG
  unsigned long long tnw_ = 0, tnq_ = 0;
#define TN_W0() tnq_ = __builtin_readcyclecounter()
#define TN_W1() tnw_ += __builtin_readcyclecounter() - tnq_
#else
#define TN_W0()
#define TN_W1()
#endif
#define LANDED(more)                                                      \
  do {                                                                    \
    TN_W0();                                                              \
    if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
    TN_W1();                                                              \
  } while (0)

  // ---- interleaved K loop (the NT kernel's default form, see gemm_nt_pipeline.h): a phase is {BARRIER, 16 MFMAs}; the
  // transposed fragment reads for the NEXT phase and the DMA issues are dealt into the shadows of those MFMAs. One barrier
  // per phase, no stagger. The register budget decides the details: with 128 accumulators, ping-pong A buffers put the
  // kernel at 254 VGPRs, i.e. two waves fill a SIMD's register file and the side stream's small kernels (16-64 VGPRs), which
  // ran BESIDE the old 223-register kernel on the same CUs, were left with the 4-13 CUs it does not use: the whole step
  // measured no faster although this kernel was 10 % faster alone. So A has ONE buffer that rolls: a phase issues its
  // MFMAs fragment-major (mi = 0..3, each 2 k-steps x 2 B fragments), and once the four MFMAs of A fragment mi have
  // issued, the reads of the NEXT A half's fragment mi go to the same registers; only fragment 3 cannot roll inside its
  // phase and is read at the start of the next one, under that phase's first 12 MFMAs, with its own lgkmcnt wait.
  // B keeps two buffers whose roles alternate from one K-tile to the next (loop written for two K-tiles).
  //  ph1 (A0,B0): reads A0[3] (late), B1(t); issue A1(t+1)        ph2 (A0,B1): reads A1(t)[0..2]; issue A0(t+2)
  //  ph3 (A1,B1): reads A1(t)[3] (late); issue B0(t+2), B1(t+2); then the vmcnt wait
  //  ph4 (A1,B0): reads A0(t+1)[0..2], B0(t+1)
  //  RAW: what ph4's shadows read (K-tile t+1's A0, B0) and what ph1 / ph2 of the next tile read (its B1, A1) was issued
  //       before the three newest half-tiles that the wait leaves in flight; a BARRIER follows the wait.
  //  WAR: a slot's previous occupant was read at the latest in the phase before the one whose shadows re-fill it, and
  //       those reads retired at that phase's BARRIER or at the explicit wait of a late fragment.
  s16x4 ra1[4][4], rb2[2][2][4];  // [A fragment][kk*2 + second], [B buffer][fragment][kk*2 + second]
// LDS addresses: one lane-constant base per (buffer, fragment) — the swizzle makes a fragment's offset non-linear in its
// index — and everything else (half-tile, image piece) in the instruction's 16-bit offset field: a buffer spans 64 KiB.
// The reads are the BUILTIN (beside builtin DMAs hipcc drains vmcnt before every LDS read that might alias one; with the
// DMAs written as asm nothing makes it do that, and
// it tracks lgkmcnt for the reads itself — as asm, their outputs looked ready at once, and the 16-bit shuffles that
// assemble a fragment from its two halves were hoisted above the arrival of the data (wrong results, not a crash).
// (The staggered two-barrier form of this loop, 6-9 % slower, was removed in round 3; it is in the history.)
#define TR1(dst, addr, OFF) dst = lds_read_tr16_addr((addr) + (OFF))
#define RD4(dst, ad_, o_)  /* the four pieces of one fragment */ \
  do { TR1(dst[0], ad_, (o_)); TR1(dst[1], ad_, (o_) + 512); TR1(dst[2], ad_, (o_) + 4096); TR1(dst[3], ad_, (o_) + 4608); PIN(); } while (0)
#define RA4(buf, h, mi) RD4(ra1[mi], abase[buf][mi], (h) * 16384)
#define RB4(bb, buf, h, ni) RD4(rb2[bb][ni], bbase[buf][ni], (2 + (h)) * 16384)
// DMA in the scalar-base form (common.h): lane-constant 32-bit offsets, the K-tile's base address on the scalar unit
#define SA_(h, kt) ((const char*)p.A + ((size_t)(t_begin + (kt) * 64) * p.lda + bn * 256 + (h) * 128) * 2)
#define SB_(h, kt) ((const char*)p.B + ((size_t)(t_begin + (kt) * 64) * p.ldb + bk * 256 + (h) * 128) * 2)
#define STG_A(buf, h, kt)                                                                     \
  do {                                                                                        \
    const char* sb_ = SA_(h, kt);                                                             \
    DMA16(sb_, vA0, ldsb + (((buf) * 4 + (h)) * HT + dst0) * 2);                              \
    DMA16(sb_, vA1, ldsb + (((buf) * 4 + (h)) * HT + dst1) * 2);                              \
    PIN();                                                                                    \
  } while (0)
#define STG_B(buf, h, kt)                                                                     \
  do {                                                                                        \
    const char* sb_ = SB_(h, kt);                                                             \
    DMA16(sb_, vB0, ldsb + (((buf) * 4 + 2 + (h)) * HT + dst0) * 2);                          \
    DMA16(sb_, vB1, ldsb + (((buf) * 4 + 2 + (h)) * HT + dst1) * 2);                          \
    PIN();                                                                                    \
  } while (0)
  const uint32_t vA0 = (uint32_t)(r0 * p.lda + sc0) * 2, vA1 = (uint32_t)(r1 * p.lda + sc1) * 2;
  const uint32_t vB0 = (uint32_t)(r0 * p.ldb + sc0) * 2, vB1 = (uint32_t)(r1 * p.ldb + sc1) * 2;
  const uint32_t ldsb = LDS_ADDR(&smem[0]);
  unsigned abase[2][4], bbase[2][2];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) abase[bf][mi] = lds0 + 65536u * bf + 2u * aoff[mi];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bbase[bf][ni] = lds0 + 65536u * bf + 2u * boff[ni];
  }
// one piece (0..3 -> byte offsets 0, 512, 4096, 4608) of an A / B fragment
#define PO_(pc) ((pc) == 0 ? 0 : (pc) == 1 ? 512 : (pc) == 2 ? 4096 : 4608)
#define RA1(buf, h, mi, pc) do { TR1(ra1[mi][pc], abase[buf][mi], (h) * 16384 + PO_(pc)); PIN(); } while (0)
#define RB1(bb, buf, h, ni, pc) do { TR1(rb2[bb][ni][pc], bbase[buf][ni], (2 + (h)) * 16384 + PO_(pc)); PIN(); } while (0)
// MFMA j of a phase, fragment-major: mi = j >> 2, kk = (j >> 1) & 1, ni = j & 1 (each accumulator still sees kk = 0 first)
#define MF1(mh, nh, bb, j)                                                                                      \
  do {                                                                                                          \
    acc[mh][(j) >> 2][nh][(j) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                   \
        FRAG(rb2[bb][(j) & 1], ((j) >> 1) & 1), FRAG(ra1[(j) >> 2], ((j) >> 1) & 1), acc[mh][(j) >> 2][nh][(j) & 1], 0, 0, 0); \
    PIN();                                                                                                      \
  } while (0)
#define PH1(mh, nh, bb, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                    \
  do {                                                                                                          \
    BARRIER(); PIN();                                                                                           \
    MF1(mh, nh, bb, 0); s0; MF1(mh, nh, bb, 1); s1; MF1(mh, nh, bb, 2); s2; MF1(mh, nh, bb, 3); s3;             \
    MF1(mh, nh, bb, 4); s4; MF1(mh, nh, bb, 5); s5; MF1(mh, nh, bb, 6); s6; MF1(mh, nh, bb, 7); s7;             \
    MF1(mh, nh, bb, 8); s8; MF1(mh, nh, bb, 9); s9; MF1(mh, nh, bb, 10); s10; MF1(mh, nh, bb, 11); s11;         \
    MF1(mh, nh, bb, 12); s12; MF1(mh, nh, bb, 13); s13; MF1(mh, nh, bb, 14); s14; MF1(mh, nh, bb, 15); s15;     \
  } while (0)
#define NOP_ (void)0
#define TWO(x, y) do { x; y; } while (0)
#define TN_STEP(bb)                                                                                             \
  do {                                                                                                          \
    constexpr int b = (bb); /* the loop alternates TN_STEP(0) / TN_STEP(1) from t = 0: buffer t & 1 is a literal */ \
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;                                                                \
    /* ph1: A0 (fragment 3 arrives under the first 12 MFMAs), B0; B1(t) for ph2 */                              \
    PH1(0, 0, bb, RA1(b, 0, 3, 0), RA1(b, 0, 3, 1), RA1(b, 0, 3, 2), RA1(b, 0, 3, 3),                           \
        RB1(1 - bb, b, 1, 0, 0), RB1(1 - bb, b, 1, 0, 1), RB1(1 - bb, b, 1, 0, 2), RB1(1 - bb, b, 1, 0, 3),     \
        RB1(1 - bb, b, 1, 1, 0), RB1(1 - bb, b, 1, 1, 1), RB1(1 - bb, b, 1, 1, 2), RB1(1 - bb, b, 1, 1, 3),     \
        TWO(if (n1) STG_A(b ^ 1, 1, t + 1), NOP_), NOP_, NOP_, NOP_);                                           \
    /* ph2: A0, B1; A1's fragments roll in behind the MFMAs that are done with A0's */                          \
    PH1(0, 1, 1 - bb, NOP_, NOP_, NOP_, NOP_, RA1(b, 1, 0, 0), RA1(b, 1, 0, 1), RA1(b, 1, 0, 2), RA1(b, 1, 0, 3), \
        RA1(b, 1, 1, 0), RA1(b, 1, 1, 1), RA1(b, 1, 1, 2), RA1(b, 1, 1, 3),                                     \
        RA1(b, 1, 2, 0), RA1(b, 1, 2, 1), RA1(b, 1, 2, 2), TWO(RA1(b, 1, 2, 3), TWO(if (n2) STG_A(b, 0, t + 2), NOP_))); \
    /* ph3: A1 (fragment 3 late), B1 */                                                                         \
    PH1(1, 1, 1 - bb, RA1(b, 1, 3, 0), RA1(b, 1, 3, 1), RA1(b, 1, 3, 2), RA1(b, 1, 3, 3),                       \
        TWO(if (n2) STG_B(b, 0, t + 2), NOP_), NOP_, NOP_, NOP_, TWO(if (n2) STG_B(b, 1, t + 2), NOP_), NOP_, NOP_, NOP_, \
        NOP_, NOP_, NOP_, NOP_);                                                                                \
    LANDED(n2);                                                                                                 \
    /* ph4: A1, B0; the next K-tile's A0[0..2] roll in, its B0 goes to the buffer B1 has left */                \
    PH1(1, 0, bb, RB1(1 - bb, b ^ 1, 0, 0, 0), RB1(1 - bb, b ^ 1, 0, 0, 1), RB1(1 - bb, b ^ 1, 0, 0, 2),        \
        RB1(1 - bb, b ^ 1, 0, 0, 3),                                                                            \
        TWO(RA1(b ^ 1, 0, 0, 0), RB1(1 - bb, b ^ 1, 0, 1, 0)), TWO(RA1(b ^ 1, 0, 0, 1), RB1(1 - bb, b ^ 1, 0, 1, 1)), \
        TWO(RA1(b ^ 1, 0, 0, 2), RB1(1 - bb, b ^ 1, 0, 1, 2)), TWO(RA1(b ^ 1, 0, 0, 3), RB1(1 - bb, b ^ 1, 0, 1, 3)), \
        RA1(b ^ 1, 0, 1, 0), RA1(b ^ 1, 0, 1, 1), RA1(b ^ 1, 0, 1, 2), RA1(b ^ 1, 0, 1, 3),                     \
        RA1(b ^ 1, 0, 2, 0), RA1(b ^ 1, 0, 2, 1), RA1(b ^ 1, 0, 2, 2), RA1(b ^ 1, 0, 2, 3));                    \
  } while (0)
  if (nk > 0) {
    STG_A(0, 0, 0); STG_B(0, 0, 0); STG_B(0, 1, 0); STG_A(0, 1, 0);
    if (nk > 1) { STG_A(1, 0, 1); STG_B(1, 0, 1); STG_B(1, 1, 1); }
    LANDED(nk > 1);
  }
  BARRIER();
#if TN_DBG
  tnw_ = 0;
  const unsigned long long tn_c0 = __builtin_readcyclecounter(), tn_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (nk > 0) {
    RA4(0, 0, 0); RA4(0, 0, 1); RA4(0, 0, 2); RB4(0, 0, 0, 0); RB4(0, 0, 0, 1);
    int t = 0;
    while (true) {
      TN_STEP(0);
      if (++t >= nk) break;
      TN_STEP(1);
      if (++t >= nk) break;
    }
    BARRIER();
  }
#if TN_DBG
  if ((blockIdx.x == 17 || blockIdx.x == 130) && lane == 0 && (uw == 0 || uw == 5)) {
    const unsigned long long dc = __builtin_readcyclecounter() - tn_c0, dr = __builtin_amdgcn_s_memrealtime() - tn_r0;
    printf("TN blk %d wave %d: %d K-tiles, %.3f us per K-tile, %.0f MHz, vmcnt wait %.1f %% of the loop\n", (int)blockIdx.x, uw, nk,
           (double)dr / 100.0 / nk, (double)dc / ((double)dr / 100.0), 100.0 * (double)tnw_ / (double)dc);
  }
#endif
#undef TN_STEP
#undef TWO
#undef NOP_
#undef PH1
#undef MF1
#undef RA1
#undef RB1
#undef PO_
#undef RA4
#undef RB4
#undef RD4
#undef TR1
#undef STG_A
#undef STG_B
#undef SA_
#undef SB_
#undef FRAG
#undef LANDED

  // D[row = k][col = n] (B fragment first): lane owns dW[n = .. + li][k0 .. k0+3]
  float* out = p.slab + (size_t)split * p.N * p.K;
  // the lane's coordinates are re-derived here (from mbcnt, behind an opaque copy): carried across the K loop they — or the
  // thread id they come from — cost the registers that decide whether the kernel fits 224 VGPRs
  int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));  // the lane id without v0
  asm volatile("" : "+v"(lane_e));
  const int li_e = lane_e & 15, fg_e = lane_e >> 4;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int n = bn * 256 + mh * 128 + wm * 64 + mi * 16 + li_e;
      if (n >= p.N) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int k0 = bk * 256 + nh * 128 + wn * 32 + ni * 16 + 4 * fg_e;
          const f32x4 v = acc[mh][mi][nh][ni];
          *(float4*)(out + (size_t)n * p.K + k0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

// Token-major weight gradient on the big-tile pipeline: Ncols % 256 == 0 (readable columns of A), K % 256
// == 0, rows_per_split % 64 == 0. N (rows stored) may be smaller than Ncols.
extern "C" int plb_launch_gemm_tn_big(const PlbGemmTN* p, hipStream_t stream) {
  if (p->Ncols % 256 || p->K % 256 || p->rows_per_split % 64 || p->Mtot % 64 || p->splits <= 0) return 1;
  if ((long)p->splits * p->rows_per_split < p->Mtot) return 1;
  dim3 grid((p->Ncols / 256) * (p->K / 256) * p->splits), block(512);
  hipLaunchKernelGGL(gemm_tn_big_kernel, grid, block, 0, stream, *p);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// K-loop form: 1 interleaved (reads / DMA issues in the MFMA shadows), 0 staggered two-barrier loop,
// -1 (default) per tile: measured on 16384 x {768, 2048, 2304} x {768, 2048, 2304}, interleaved wins by
// 5-8 % on 128x384 and 128x256, staggered by 5-10 % on a plain 256x256 launch (whose phase 3 has nothing to
// interleave and phase 4 twelve reads) — but with the GELU epilogues (the only 256x256 launches of the model:
// 16384 x 2048 x 768) interleaved wins again by 6 %: 68.7 vs 72.8 us forward, 70.0 vs 74.7 us backward.
static int g_nt_prefetch = -1;
extern "C" void plb_set_gemm_nt_prefetch(int on) { g_nt_prefetch = on; }

// tile = 256: 256x256 (M % 256, N % 256); 384: 128x384 (M % 128, N % 384); 1256: 128x256 (M % 128, N % 256).
extern "C" int plb_launch_gemm_nt_big(const PlbGemmNT* p, int tile, int act, int out_f32, hipStream_t stream) {
  const bool pf = g_nt_prefetch < 0 ? (tile != 256 || act != 0) : (g_nt_prefetch != 0);
  if (pf)
    return tile == 384 ? launch_big<3, true>(p, act, out_f32, stream)
           : tile == 1256 ? launch_big<1, true>(p, act, out_f32, stream) : launch_big<2, true>(p, act, out_f32, stream);
  return tile == 384 ? launch_big<3, false>(p, act, out_f32, stream)
         : tile == 1256 ? launch_big<1, false>(p, act, out_f32, stream) : launch_big<2, false>(p, act, out_f32, stream);
}
