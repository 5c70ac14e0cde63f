// dW[N,K] = A[rows,N]^T · B[rows,K] on 1-byte operand images (gfx950): the token-major ("TN") weight-gradient GEMM of
// gemm_big.hip in fp8 mode (BASELINE.json configs[4]). A = the e5m2 image of a gradient (dQKV, dpre1, dU, dpre2), B = the
// e4m3 image of the activation it multiplies (x, context, a, gelu(u)), both written by the launches that produced the
// tensors (engine.cpp), fp32 accumulation, fp32 slabs: the reference's dW = dY^T·X of autograd (train.py:356), summed over
// the 12 applications of the shared layer in ONE launch per weight.
//
// Same pipeline as gemm_tn_big_kernel — 256 (n) x 256 (k) output tile, one 8-wave workgroup per CU, LDS-DMA in full
// 128-byte lines into a double-buffered 128 KiB stage, interleaved K loop with one barrier per phase — on twice the tokens
// per byte: a half-tile is [128 t][128 columns] bytes (16 KiB), a K-tile 128 tokens, and ONE block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales (plain fp8 arithmetic at twice the bf16 MFMA rate) replaces four
// bf16 MFMAs. MFMA fragments need the reduction index (the token) on the k axis, i.e. COLUMNS of the token-major images:
// ds_read_b64_tr_b8. Measured semantics (tools/probe_tr8.hip, profiles/r04_probe_tr8_hw.txt): within a 16-lane group,
// lane 2q + p supplies the address of row q (0..7), 8-byte column piece p (0..1); lane i receives column i's 8 rows.
// A fragment = 32 tokens per lane = 4 such reads; lane group g = lane >> 4 takes tokens [32 g, 32 g + 32) of the K-tile,
// read r the rows 32 g + 8 r + {0..7}. A and B use the same assignment, so the k order inside the MFMA does not matter.
// Banks: a half-wave's read touches 2 groups x 8 rows x 16 B; a 128-byte row is 32 banks, so the 8 rows of one parity
// must fall into 8 different 16-byte units: unit' = unit ^ f(row), f(row) = ((row >> 1) & 3) | (((row >> 5) & 1) << 2)
// (rows of one group differ in bits 1-2, the two groups of a half-wave in bit 5), applied to the DMA's per-lane SOURCE unit.
#include "gemm_nt_pipeline.h"

namespace {

typedef int i32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) i32x2v lds_i32x2v;
DEVI i32x2v lds_read_tr8_addr(uint32_t lds_byte_addr) {
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2v*)(uintptr_t)lds_byte_addr);
}
DEVI int tn8_f(int row) { return ((row >> 1) & 3) | (((row >> 5) & 1) << 2); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(224))) void gemm_tn_fp8_kernel(PlbGemmTN p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * 16384];  // [buf][A0,A1,B0,B1][128 t][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = uw >> 2, wn = uw & 3;
  const int nbk = p.K >> 8;
  const int tiles = (p.Ncols >> 8) * nbk;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int split = logical / tiles, tile = logical % tiles;
  const int bn = tile / nbk, bk = tile % nbk;
  const int t_begin = split * p.rows_per_split;
  int t_end = t_begin + p.rows_per_split;
  if (t_end > p.Mtot) t_end = p.Mtot;
  const int nk = (t_end - t_begin) >> 7;

  // ---- staging: instruction i = 2w + j of a half-tile covers rows 8 i + (lane >> 3); lane writes physical unit lane & 7
  const int r0 = (2 * uw) * 8 + (lane >> 3), r1 = (2 * uw + 1) * 8 + (lane >> 3);
  const int sc0 = ((lane & 7) ^ tn8_f(r0)) * 16, sc1 = ((lane & 7) ^ tn8_f(r1)) * 16;   // source column (bytes) within the half
  const int dst0 = (2 * uw) * 1024, dst1 = (2 * uw + 1) * 1024;
  // ---- fragment reads: lane (g, q, pp) of read r supplies row 32 g + 8 r + q, unit (logical ^ f(row)), piece pp
  const int fg = lane >> 4, li = lane & 15, q8 = li >> 1, pp = li & 1;
  const int fx = tn8_f(32 * fg + q8);                  // (8 r never touches bits 1, 2, 5)
  const int rowbase = (32 * fg + q8) * 128 + pp * 8;
  const unsigned lds0 = (unsigned)(size_t)&smem[0];

  f32x4 acc[2][4][2][2];  // [mh][mi][nh][ni]: n = mh*128 + wm*64 + mi*16 + .., k = nh*128 + wn*32 + ni*16 + ..
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
#define LANDED(more)                                                      \
  do {                                                                    \
    if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
  } while (0)

  // Fragments as 8-register tuples written piece by piece (sub-registers: no copy to form the MFMA operand). One rolling
  // A buffer, two B buffers — the schedule of gemm_tn_big_kernel, slot for slot: a phase is {BARRIER, 16 slots}; the fp8
  // MFMA of (mi, ni) sits in slot 4 mi + 2 ni (8 MFMAs of twice the length), the transposed reads for the next phase
  // and the DMA issues are dealt into the same slots as in the bf16 kernel (a fragment is 4 pieces there as here).
  i32x8v ra[4], rb[2][2];
#define TR1(dst, pc, addr, OFF)                                         \
  do {                                                                  \
    const i32x2v t_ = lds_read_tr8_addr((addr) + (OFF));                \
    dst[2 * (pc)] = t_[0]; dst[2 * (pc) + 1] = t_[1];                   \
  } while (0)
#define SA_(h, kt) ((const char*)p.A + ((size_t)(t_begin + (kt) * 128) * p.lda + bn * 256 + (h) * 128))
#define SB_(h, kt) ((const char*)p.B + ((size_t)(t_begin + (kt) * 128) * p.ldb + bk * 256 + (h) * 128))
#define STG_A(buf, h, kt)                                                                     \
  do {                                                                                        \
    const char* sb_ = SA_(h, kt);                                                             \
    DMA16(sb_, vA0, ldsb + ((buf) * 4 + (h)) * 16384 + dst0);                                 \
    DMA16(sb_, vA1, ldsb + ((buf) * 4 + (h)) * 16384 + dst1);                                 \
    PIN();                                                                                    \
  } while (0)
#define STG_B(buf, h, kt)                                                                     \
  do {                                                                                        \
    const char* sb_ = SB_(h, kt);                                                             \
    DMA16(sb_, vB0, ldsb + ((buf) * 4 + 2 + (h)) * 16384 + dst0);                             \
    DMA16(sb_, vB1, ldsb + ((buf) * 4 + 2 + (h)) * 16384 + dst1);                             \
    PIN();                                                                                    \
  } while (0)
  const uint32_t vA0 = (uint32_t)(r0 * p.lda + sc0), vA1 = (uint32_t)(r1 * p.lda + sc1);
  const uint32_t vB0 = (uint32_t)(r0 * p.ldb + sc0), vB1 = (uint32_t)(r1 * p.ldb + sc1);
  const uint32_t ldsb = LDS_ADDR(&smem[0]);
  // one lane-constant base per (buffer, fragment): the swizzle makes a fragment's offset non-linear in its index;
  // half-tile and piece go into the instruction's 16-bit offset field (a buffer spans 64 KiB)
  unsigned abase[2][4], bbase[2][2];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) abase[bf][mi] = lds0 + 65536u * bf + rowbase + (((wm * 4 + mi) ^ fx) << 4);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bbase[bf][ni] = lds0 + 65536u * bf + rowbase + (((wn * 2 + ni) ^ fx) << 4);
  }
// piece pc (0..3) of a fragment = rows + 8 pc: byte offset 1024 pc
#define RA1(buf, h, mi, pc) do { TR1(ra[mi], pc, abase[buf][mi], (h) * 16384 + (pc) * 1024); PIN(); } while (0)
#define RB1(bb, buf, h, ni, pc) do { TR1(rb[bb][ni], pc, bbase[buf][ni], (2 + (h)) * 16384 + (pc) * 1024); PIN(); } while (0)
#define RA4(buf, h, mi) do { RA1(buf, h, mi, 0); RA1(buf, h, mi, 1); RA1(buf, h, mi, 2); RA1(buf, h, mi, 3); } while (0)
#define RB4(bb, buf, h, ni) do { RB1(bb, buf, h, ni, 0); RB1(bb, buf, h, ni, 1); RB1(bb, buf, h, ni, 2); RB1(bb, buf, h, ni, 3); } while (0)
// slot j of a phase: the MFMA of (mi = j >> 2, ni = (j >> 1) & 1) in the even slots. First operand (rows of D) = the
// activation fragment (e4m3), second = the gradient fragment (e5m2): D[row = k][col = n], as in the bf16 kernel.
#define MF1(mh, nh, bb, j)                                                                                      \
  do {                                                                                                          \
    if (((j) & 1) == 0)                                                                                         \
      acc[mh][(j) >> 2][nh][((j) >> 1) & 1] =                                                                   \
          mfma_fp8<true>(rb[bb][((j) >> 1) & 1], ra[(j) >> 2], acc[mh][(j) >> 2][nh][((j) >> 1) & 1]);          \
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
// The phase table of gemm_tn_big_kernel (see there for the RAW / WAR argument; fragment mi's last MFMA is in slot
// 4 mi + 2, so a read into ra[mi] in slot 4 mi + 4 or later follows it):
//  ph1 (A0,B0): reads A0[3] (late), B1(t); issue A1(t+1)        ph2 (A0,B1): reads A1(t)[0..2]; issue A0(t+2)
//  ph3 (A1,B1): reads A1(t)[3] (late); issue B0(t+2), B1(t+2); then the vmcnt wait
//  ph4 (A1,B0): reads A0(t+1)[0..2], B0(t+1)
#define TN_STEP(bb)                                                                                             \
  do {                                                                                                          \
    constexpr int b = (bb);                                                                                     \
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;                                                                \
    PH1(0, 0, bb, RA1(b, 0, 3, 0), RA1(b, 0, 3, 1), RA1(b, 0, 3, 2), RA1(b, 0, 3, 3),                           \
        RB1(1 - bb, b, 1, 0, 0), RB1(1 - bb, b, 1, 0, 1), RB1(1 - bb, b, 1, 0, 2), RB1(1 - bb, b, 1, 0, 3),     \
        RB1(1 - bb, b, 1, 1, 0), RB1(1 - bb, b, 1, 1, 1), RB1(1 - bb, b, 1, 1, 2), RB1(1 - bb, b, 1, 1, 3),     \
        TWO(if (n1) STG_A(b ^ 1, 1, t + 1), NOP_), NOP_, NOP_, NOP_);                                           \
    PH1(0, 1, 1 - bb, NOP_, NOP_, NOP_, NOP_, RA1(b, 1, 0, 0), RA1(b, 1, 0, 1), RA1(b, 1, 0, 2), RA1(b, 1, 0, 3), \
        RA1(b, 1, 1, 0), RA1(b, 1, 1, 1), RA1(b, 1, 1, 2), RA1(b, 1, 1, 3),                                     \
        RA1(b, 1, 2, 0), RA1(b, 1, 2, 1), RA1(b, 1, 2, 2), TWO(RA1(b, 1, 2, 3), TWO(if (n2) STG_A(b, 0, t + 2), NOP_))); \
    PH1(1, 1, 1 - bb, RA1(b, 1, 3, 0), RA1(b, 1, 3, 1), RA1(b, 1, 3, 2), RA1(b, 1, 3, 3),                       \
        TWO(if (n2) STG_B(b, 0, t + 2), NOP_), NOP_, NOP_, NOP_, TWO(if (n2) STG_B(b, 1, t + 2), NOP_), NOP_, NOP_, NOP_, \
        NOP_, NOP_, NOP_, NOP_);                                                                                \
    LANDED(n2);                                                                                                 \
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
#undef TN_STEP
#undef TWO
#undef NOP_
#undef PH1
#undef MF1
#undef RA1
#undef RB1
#undef RA4
#undef RB4
#undef TR1
#undef STG_A
#undef STG_B
#undef SA_
#undef SB_
#undef LANDED

  // D[row = k][col = n]: lane owns dW[n = .. + li][k0 .. k0+3]; both per-tensor dequantisation factors applied here
  const float deq = p.deq_a[0] * p.deq_b[0];
  float* out = p.slab + (size_t)split * p.N * p.K;
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
          *(float4*)(out + (size_t)n * p.K + k0) = make_float4(v[0] * deq, v[1] * deq, v[2] * deq, v[3] * deq);
        }
    }
}

}  // namespace

// Ncols % 256 == 0 (readable byte columns of A), K % 256 == 0, rows_per_split % 128 == 0, Mtot % 128 == 0; lda / ldb in
// bytes, multiples of 16; deq_a / deq_b: device scalars (1 / scale of each image).
extern "C" int plb_launch_gemm_tn_fp8(const PlbGemmTN* p, hipStream_t stream) {
  if (p->Ncols % 256 || p->K % 256 || p->rows_per_split % 128 || p->Mtot % 128 || p->splits <= 0) return 1;
  if ((long)p->splits * p->rows_per_split < p->Mtot) return 1;
  if (p->lda % 16 || p->ldb % 16 || !p->deq_a || !p->deq_b) return 1;
  dim3 grid((p->Ncols / 256) * (p->K / 256) * p->splits), block(512);
  hipLaunchKernelGGL(gemm_tn_fp8_kernel, grid, block, 0, stream, *p);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
