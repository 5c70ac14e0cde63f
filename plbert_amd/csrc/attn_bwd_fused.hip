// Single-kernel attention backward for sequences of at most 512 keys, head_dim 64, gfx950.
// Gradient of AlbertAttention's softmax(q·k^T·d^-0.5 + key_padding_mask)·v (modeling_albert.py:110-135, 166-200) with
// FIVE products per (query block, key block) — S, dP, dV^T, dK^T, dQ^T — and ONE exponential per score; the two-kernel
// form in attn.hip recomputes S and dP in both kernels (seven products, two exponential passes) and remains the path for
// longer sequences.
//
// One workgroup = 4 waves = one (batch, head); ONE wave per SIMD with the whole 512-entry register file:
//   * wave w OWNS keys [128 w, 128 w + 128): their K and V row fragments stay in registers (128 VGPRs) and so do
//     dK^T[64 d][128 keys] and dV^T[64 dv][128 keys] (256 accumulator registers), so dK and dV are never summed across
//     waves or workgroups — S <= 512 is what makes one workgroup own EVERY key of its (batch, head);
//   * the workgroup walks the query blocks of 32. Per block a wave computes, for each of its four 32-key blocks,
//     S' = Q·K^T - LSE and dP' = dO·V^T - delta with the key on the MFMA lane (row constants as the initial accumulators),
//     P = exp2(c·S'), dV^T += dO^T·P and dK^T += Q^T·dS (dS = P∘dP') with the P / dS accumulators taken directly as B
//     operands, and writes dS (bf16) into an LDS exchange image [512 keys][32 q];
//   * dQ of the PREVIOUS block is computed from that image one barrier later, beside this block's products:
//     dQ^T[64 d][32 q] = K^T[64 d][512 keys]·dS^T[512 keys][32 q] as 4 x 2 tiles of 16x16 (MFMA 16x16x32), wave w taking
//     d rows [16 w, 16 w + 16) for both query halves over ALL keys — no partial sums, no atomics, nothing summed in HBM:
//     dQ is complete in one accumulator and bitwise reproducible. K^T comes from a [512 keys][64 d] LDS image (filled once
//     by LDS-DMA), dS^T from the exchange image, both by ds_read_b64_tr_b16 on swizzles that make the reads conflict-free.
// One barrier per query block: Q / dO tiles (one dual-use image each: row reads for S / dP, transposed reads for dV^T /
// dK^T) arrive by LDS-DMA a block ahead; delta = rowsum(dO∘O) and LSE of the next block are formed from plain loads
// meanwhile. LDS: K image 64 KiB + 2 x 32 KiB exchange + 2 x 8 KiB staging + statistics = 144.5 KiB.
#include "attn_common.h"

namespace {

constexpr int FB_KIMG = 0;                  // [512 keys][128 B], 32-B quarter q of a row stored at q ^ kt(key)
constexpr int FB_EXCH = 65536;              // 2 x [512 keys][64 B], 8-B unit u (4 queries) of a row stored at u ^ ((key>>1)&7)
constexpr int FB_STG = FB_EXCH + 65536;     // 2 stages x (Q | dO), each [32 q][128 B] dual-use image
constexpr int FB_STAT = FB_STG + 16384;     // 2 stages x (lse[32] | delta[32]) fp32
constexpr int FB_TOTAL = FB_STAT + 512;

// K image: a half-wave's transposed read of a 16x16x32 operand touches keys {k..k+3, k+8..k+11} x 32 B; with the quarter
// XOR-ed by ((key>>1)&1) | ((key>>3)&1)<<1 those eight segments cover the 64 banks exactly once.
DEVI int kt_of(int key) { return ((key >> 1) & 1) | (((key >> 3) & 1) << 1); }
// Dual-use [32 rows][64 cols] image with 128-B rows: 16-B chunk c of row r stored at c ^ du_f(r). The 8 same-parity rows
// of a ds_read_b128 lane group get 8 different values (conflict-free row reads of the 32x32x16 A operand), and rows r,
// r + 2 differ in bit 2, i.e. in the 64-B half a transposed read's four rows take (conflict-free ds_read_b64_tr_b16).
DEVI int du_f(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1); }

typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

DEVI bf16x8 join_tr(s16x4 a, s16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
// LDS accesses by BYTE ADDRESS: every address below is a lane-constant base register (made opaque once, so that hipcc
// keeps it as a base) plus a compile-time offset that fits the instructions' 16-bit offset field. Written as smem + region
// + stage + lane offset, hipcc folded the constants past 64 KiB into one VGPR per (region, stage, step): 64 address
// registers for the dQ phase alone, and the kernel spilled.
typedef __attribute__((address_space(3))) const bf16x8 lds_cbf16x8;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
typedef __attribute__((address_space(3))) u32x2_t lds_u32x2;
DEVI bf16x8 lds_rd128(uint32_t a) { return *(lds_cbf16x8*)(uintptr_t)a; }
DEVI f32x4 lds_rdf4(uint32_t a) { return *(lds_cf32x4*)(uintptr_t)a; }
DEVI void lds_wr64(uint32_t a, uint32_t x, uint32_t y) { *(lds_u32x2*)(uintptr_t)a = u32x2_t{x, y}; }
DEVI bf16x8 lds_tr2(uint32_t a0, uint32_t a1) { return join_tr(lds_read_tr16_addr(a0), lds_read_tr16_addr(a1)); }
#define OPAQUE(x) asm volatile("" : "+v"(x))

#ifndef FUSED_DBG
#define FUSED_DBG 0   // timing builds: 1 no exponentials, 2 no MFMA (attn_common.h: ATTN_DBG); 256 no dQ phase, 512 no key-owner phase
#endif

// F8: the call also (or only) writes the e5m2 image of dQKV (fp8 mode); a separate instantiation, so that the bf16 build
// does not carry the image's scale, maximum and pointers through a loop that runs at the full register file
template <int MODE>   // 0: bf16 rows, 1: the e5m2 image alone (fp8 training call), 2: both
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_kernel(PlbAttn p) {
  constexpr bool F8 = MODE != 0, B16 = MODE != 1;
  __shared__ __attribute__((aligned(16))) unsigned char smem[FB_TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hd = blockIdx.x % p.NH, b = blockIdx.x / p.NH;
  const int S = p.S, H = p.H;
  int len = p.lengths ? p.lengths[b] : S;
  len = len < 1 ? 1 : (len > S ? S : len);
  const size_t tok0 = (size_t)b * S;
  const int ld = p.ldqkv, ldo = p.lddctx;
  const int lk = lane & 31, h = lane >> 5;
  const float sl2 = p.scale * LOG2E;
  const uint32_t lds_base = LDS_ADDR(smem);
  const int QT = (S + 127) >> 7;

  // ---- K image: wave w fills the rows of its own keys (every row below 32*ceil(len/32) is read by the dQ phase; rows past
  // S repeat row S-1: finite values that only ever meet dS = 0)
  {
    const char* gk = (const char*)(p.qkv + H + hd * 64 + tok0 * ld);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = wave * 128 + 8 * j + (lane >> 3);
      const int rc = row < S ? row : S - 1;
      const int chunk = ((((lane & 7) >> 1) ^ kt_of(row)) << 1) | (lane & 1);
      DMA16(gk, (uint32_t)(rc * ld + chunk * 8) * 2, lds_base + FB_KIMG + (wave * 128 + 8 * j) * 128);
    }
  }
  // ---- Q / dO staging of one 32-query block: waves 0,1 the Q image, waves 2,3 the dO image, 16 rows each
  const bool st_do = wave >= 2;
  const char* const st_base = st_do ? (const char*)(p.dctx + hd * 64 + tok0 * ldo) : (const char*)(p.qkv + hd * 64 + tok0 * ld);
  const int st_ld = st_do ? ldo : ld;
  uint32_t st_v0, st_v1;
  {
    const int st_r0 = 16 * (wave & 1) + (lane >> 3), st_r1 = st_r0 + 8;
    const int st_c0 = ((lane & 7) ^ du_f(st_r0)) * 8, st_c1 = ((lane & 7) ^ du_f(st_r1)) * 8;
    st_v0 = (uint32_t)(st_r0 * st_ld + st_c0) * 2; st_v1 = (uint32_t)(st_r1 * st_ld + st_c1) * 2;
  }
  const uint32_t st_lds = lds_base + FB_STG + (wave >> 1) * 4096 + (wave & 1) * 2048;
  // Lane constants that only a rare path or the epilogue needs are NOT kept across the query-block loop (the loop runs at the
  // full 512 registers: carried along they were 21 spilled registers, reloaded from scratch inside it). They are
  // re-derived where used from the lane id (mbcnt) behind an opaque copy, which keeps hipcc from hoisting them back.
#define LANE_ID(x) int x = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); OPAQUE(x)
#define F_STAGE(ST, qb_)                                                                              \
  do {                                                                                                \
    const int q0_ = (qb_) * 32;                                                                       \
    const uint32_t l_ = st_lds + (ST) * 8192;                                                         \
    if (q0_ + 32 <= S) {                                                                              \
      const char* sb_ = st_base + (size_t)q0_ * st_ld * 2;                                            \
      DMA16(sb_, st_v0, l_); DMA16(sb_, st_v1, l_ + 1024);                                            \
    } else { /* the block that crosses S: rows are clamped, every lane computes its own offsets */    \
      LANE_ID(ln_);                                                                                   \
      const int r0_ = 16 * (wave & 1) + (ln_ >> 3), r1_ = r0_ + 8;                                    \
      const int c0_ = ((ln_ & 7) ^ du_f(r0_)) * 8, c1_ = ((ln_ & 7) ^ du_f(r1_)) * 8;                 \
      const int a0_ = min(q0_ + r0_, S - 1), a1_ = min(q0_ + r1_, S - 1);                             \
      DMA16(st_base, (uint32_t)(a0_ * st_ld + c0_) * 2, l_);                                          \
      DMA16(st_base, (uint32_t)(a1_ * st_ld + c1_) * 2, l_ + 1024);                                   \
    }                                                                                                 \
  } while (0)
  // ---- row statistics of one block: thread t takes row t>>3, 8 columns; delta = rowsum(dO∘O), LSE copied
  // (wave-uniform bases + 32-bit lane offsets: one address register per load instead of a 64-bit pointer per tensor)
  const bf16_t* const g_do = p.dctx + hd * 64 + tok0 * ldo;
  const bf16_t* const g_o = p.ctx + hd * 64 + tok0 * p.ldctx;
  const float* const g_lse = p.lse + ((size_t)b * p.NH + hd) * S;
  uint4 sv_do, sv_o;
  float sv_lse = 0.f;
#define F_STAT_LOAD(qb_)                                                                              \
  do {                                                                                                \
    const int r_ = min((qb_) * 32 + (tid >> 3), S - 1);                                               \
    sv_do = *(const uint4*)(g_do + (uint32_t)(r_ * ldo + (tid & 7) * 8));                             \
    sv_o = *(const uint4*)(g_o + (uint32_t)(r_ * p.ldctx + (tid & 7) * 8));                           \
    if (tid < 32) sv_lse = g_lse[(uint32_t)min((qb_) * 32 + tid, S - 1)];                             \
  } while (0)
#define F_STAT_STORE(ST, qb_)                                                                         \
  do {                                                                                                \
    float d_ = bf_lo(sv_do.x) * bf_lo(sv_o.x) + bf_hi(sv_do.x) * bf_hi(sv_o.x);                       \
    d_ += bf_lo(sv_do.y) * bf_lo(sv_o.y) + bf_hi(sv_do.y) * bf_hi(sv_o.y);                            \
    d_ += bf_lo(sv_do.z) * bf_lo(sv_o.z) + bf_hi(sv_do.z) * bf_hi(sv_o.z);                            \
    d_ += bf_lo(sv_do.w) * bf_lo(sv_o.w) + bf_hi(sv_do.w) * bf_hi(sv_o.w);                            \
    d_ += __shfl_xor(d_, 1, 64); d_ += __shfl_xor(d_, 2, 64); d_ += __shfl_xor(d_, 4, 64);            \
    int t_ = tid; OPAQUE(t_);   /* the two LDS addresses are formed here, not carried across the loop */ \
    float* st_ = (float*)(smem + FB_STAT + (ST) * 256);  /* [bias 32 | -delta 32] */                    \
    if ((t_ & 7) == 0) st_[32 + (t_ >> 3)] = -d_;                                                     \
    /* log2-domain bias of the row: P = exp2(S*c + bias); a query past the length gets P = 0 */       \
    if (t_ < 32) st_[t_] = ((qb_) * 32 + t_ < len) ? sv_lse * sl2 : -1e30f;                           \
  } while (0)

  const int NQ = (len + 31) >> 5;   // query blocks with work (queries past the length carry exactly zero dO in this model)
  F_STAGE(0, 0);
  F_STAT_LOAD(0);

  // ---- this wave's keys: the V row fragments (B operand of dP = dO·V^T) stay in registers; the K row fragments (B
  // operand of S = Q·K^T) are re-read from the K image for every query block (2-way bank conflict: the image's swizzle
  // serves the transposed reads) — 256 accumulators + 128 fragment registers + the working set of a block do not fit
  // the 512-entry file, and hipcc then spilled ~270 registers into the loop.
  bf16x8 vf[4][4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int key = wave * 128 + kb * 32 + lk;
    const int kr = key < S ? key : S - 1;
    const bf16_t* vp = p.qkv + 2 * H + hd * 64 + (tok0 + kr) * ld + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[kb][ks] = *(const bf16x8*)(vp + ks * 16);
  }
  f32x16 dk[4][2], dv[4][2];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) { dk[kb][cb] = zero16(); dv[kb][cb] = zero16(); }

  // ---- lane constants of the LDS reads and writes: absolute LDS byte addresses of (region, stage 0, step 0)
  // row reads of a dual-use image: row lane&31, chunk 2ks + h
  uint32_t rowo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    rowo[ks] = lds_base + FB_STG + lk * 128 + (((2 * ks + h) ^ du_f(lk)) << 4);
    OPAQUE(rowo[ks]);
  }
  // transposed reads of a dual-use image (A operand X^T of a 32x32x16 MFMA, k order of an accumulator-as-operand):
  // lane (g = lane>>4, q4, p4): rows 4h + q4 (+8: second read) of the 16-row k-step, columns cb*32 + 16(g&1) + 4p4..
  uint32_t tro[2][2];
  {
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int sec = 0; sec < 2; ++sec) {
        const int row = 4 * h + q4 + 8 * sec, chunk = 4 * cb + 2 * (g & 1) + (p4 >> 1);
        tro[cb][sec] = lds_base + FB_STG + row * 128 + ((chunk ^ du_f(row)) << 4) + (p4 & 1) * 8;
        OPAQUE(tro[cb][sec]);
      }
  }
  // K row fragments out of the K image: key 128w + 32kb + lane&31, chunk 2ks + h = quarter ks, half h
  uint32_t kro[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kro[ks] = lds_base + FB_KIMG + (wave * 128 + lk) * 128 + ((ks ^ kt_of(lk)) << 5) + 16 * h;
    OPAQUE(kro[ks]);
  }
  uint32_t sto = lds_base + FB_STAT + 16 * h;   // statistics: float4 at query 8rg + 4h
  OPAQUE(sto);
  // exchange writes: lane = key, registers 4g..4g+3 = queries 8g + 4h ..: unit 2g + h of the key's row
  uint32_t exo[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    exo[g4] = lds_base + FB_EXCH + (wave * 128 + lk) * 64 + (((2 * g4 + h) ^ ((lk >> 1) & 7)) << 3);
    OPAQUE(exo[g4]);
  }
  // dQ phase, A operand K^T[16 d][32 keys]: lane (g, q', p) supplies key 8g + q' (+4), d 16w + 4p
  uint32_t dqa;
  uint32_t dqb[2][2];
  {
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const int key = 8 * g + q4;
    dqa = lds_base + FB_KIMG + key * 128 + ((wave ^ kt_of(key)) << 5) + 8 * p4;
    OPAQUE(dqa);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int sec = 0; sec < 2; ++sec) {
        const int k2 = key + 4 * sec;
        dqb[qg][sec] = lds_base + FB_EXCH + k2 * 64 + (((4 * qg + p4) ^ ((k2 >> 1) & 7)) << 3);
        OPAQUE(dqb[qg][sec]);
      }
  }
  float colq[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of the dQ values this lane stored (bias gradient partial)
  // dQ rows leave from a wave-uniform base + ONE 32-bit lane offset (query lane&15 of the half, d 16w + 4(lane>>4) ..+3);
  // fp8 calls: the e5m2 image of the same values (AS ROUNDED to bf16) x the site's scale, 4 bytes per lane; the bf16 rows
  // only if somebody reads them (p.dqkv)
  bf16_t* const g_dq = B16 ? p.dqkv + tok0 * p.lddqkv + hd * 64 + 16 * wave : nullptr;
  uint8_t* const g_dq8 = F8 ? p.dqkv8 + tok0 * p.lddqkv8 + hd * 64 + 16 * wave : nullptr;
  const float qs8 = F8 ? p.dqkv_scale[0] : 1.0f;
  float amax8 = 0.f;

  F_STAT_STORE(0, 0);
  DMA_WAIT();
  __syncthreads();

  // ---- One iteration of the query-block loop, written slot by slot. A wave is alone on its SIMD, so nothing but its own
  // instruction order overlaps the MFMA pipe with the softmax arithmetic and the LDS reads: every slot below is ONE MFMA
  // followed by the VALU / LDS instructions that run in its shadow (24 issue cycles per 32x32x16 MFMA: two
  // {fma, exp} pairs, or four multiplies and a conversion), and sched_barrier (PIN) keeps hipcc from regrouping them
  // (left alone it emitted read -> wait -> MFMA chains and the MFMA time added to everything else: 4.6 us per query
  // block against 1.3 us of MFMA issue). Per 32-key block kb of the wave, software-pipelined by one product:
  //   G2  dP' = dO·V^T - delta       4 MFMA | P = exp2(S'·c + bias), scores 0-7 (S' of THIS block is complete: see G3)
  //   G3  S' of block kb+1 = Q·K^T   4 MFMA | scores 8-15; then P -> bf16
  //   G4  dV^T += dO^T·P             4 MFMA | dS = P∘dP', dS -> bf16
  //   G5  dK^T += Q^T·dS             4 MFMA | dS into the exchange image
  // with the LDS reads of each group issued one to two groups ahead. dQ of the PREVIOUS query block follows as its own
  // phase (16 key steps of 2 MFMA 16x16x32, three key steps of transposed reads in flight); all 16 key steps always
  // run: key blocks without a valid key hold dS = 0 in the exchange image.
  // Masking costs nothing in the common case: a query past the length has bias -1e30 in the statistics (P = 0), a key
  // block that CROSSES the length takes 16 selects under a wave-uniform branch.
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define TRF(imgoff, cb, s2) lds_tr2(tro[cb][0] + (imgoff) + (s2) * 2048, tro[cb][1] + (imgoff) + (s2) * 2048)
  bool key_ok[4], key_cross[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int k0 = wave * 128 + kb * 32;
    key_ok[kb] = k0 + lk < len;
    key_cross[kb] = k0 + 32 > len;   // wave-uniform: the block holds a key past the length
  }
  const bool wave_has_keys = wave * 128 < len;
  // e(r): one score -> probability; m(r): one dS
#define E_(r) s[r] = EXP2(__builtin_fmaf(s[r], sl2, ((r) < 4 ? b0 : (r) < 8 ? b1 : (r) < 12 ? b2 : b3)[(r) & 3]))
#define M_(r) dp[r] = s[r] * dp[r]
#define RD_FD(CUR) /* dO row fragments, -delta (the dP accumulator starts there) and the bias of the staged block */ \
  do {                                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fd[ks] = lds_rd128(rowo[ks] + (CUR) * 8192 + 4096);         \
    n0 = lds_rdf4(sto + (CUR) * 256 + 128); n1 = lds_rdf4(sto + (CUR) * 256 + 160);                              \
    n2 = lds_rdf4(sto + (CUR) * 256 + 192); n3 = lds_rdf4(sto + (CUR) * 256 + 224);                              \
    b0 = lds_rdf4(sto + (CUR) * 256); b1 = lds_rdf4(sto + (CUR) * 256 + 32);                                     \
    b2 = lds_rdf4(sto + (CUR) * 256 + 64); b3 = lds_rdf4(sto + (CUR) * 256 + 96);                                \
  } while (0)
  // entering: s = S' of block kb (complete), fd / n0-n3 read, fk = K rows of block kb+1 (kb < 3)
#define KO_KB(CUR, kb)                                                                                           \
  {                                                                                                              \
    constexpr int qi_ = (CUR) * 8192, di_ = (CUR) * 8192 + 4096;  /* image offsets from stage 0's Q image */     \
    f32x16 dp, sn = zero16();                                                                                    \
    bf16x8 tdo[2][2], tq[2][2];                                                                                  \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) { dp[e] = n0[e]; dp[4 + e] = n1[e]; dp[8 + e] = n2[e]; dp[12 + e] = n3[e]; } \
    /* G2 */                                                                                                     \
    dp = MFMA32(fd[0], vf[kb][0], dp); E_(0); E_(1); tdo[0][0] = TRF(di_, 0, 0); PIN();                          \
    dp = MFMA32(fd[1], vf[kb][1], dp); E_(2); E_(3); tdo[0][1] = TRF(di_, 1, 0); PIN();                          \
    dp = MFMA32(fd[2], vf[kb][2], dp); E_(4); E_(5); tdo[1][0] = TRF(di_, 0, 1); PIN();                          \
    dp = MFMA32(fd[3], vf[kb][3], dp); E_(6); E_(7); tdo[1][1] = TRF(di_, 1, 1); PIN();                          \
    /* G3 */                                                                                                     \
    if ((kb) < 3) {                                                                                              \
      sn = MFMA32(fq[0], fk[0], sn); E_(8); E_(9); PIN();                                                        \
      sn = MFMA32(fq[1], fk[1], sn); E_(10); E_(11); PIN();                                                      \
      sn = MFMA32(fq[2], fk[2], sn); E_(12); E_(13); PIN();                                                      \
      sn = MFMA32(fq[3], fk[3], sn); E_(14); E_(15); PIN();                                                      \
    } else {                                                                                                     \
      E_(8); E_(9); E_(10); E_(11); E_(12); E_(13); E_(14); E_(15); PIN();                                       \
    }                                                                                                            \
    if (key_cross[kb]) {                                                                                         \
      asm volatile("; key block crosses the length"); /* keeps the branch: hipcc if-converts it otherwise */     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) s[r] = key_ok[kb] ? s[r] : 0.f;                             \
    }                                                                                                            \
    const bf16x8 pb0 = acc_frag(s, 0);                                                                           \
    PIN();                                                                                                       \
    /* G4 */                                                                                                     \
    dv[kb][0] = MFMA32(tdo[0][0], pb0, dv[kb][0]);                                                               \
    const bf16x8 pb1 = acc_frag(s, 1); tq[0][0] = TRF(qi_, 0, 0); PIN();                                         \
    dv[kb][1] = MFMA32(tdo[0][1], pb0, dv[kb][1]); M_(0); M_(1); M_(2); M_(3); M_(4); M_(5); tq[0][1] = TRF(qi_, 1, 0); PIN(); \
    dv[kb][0] = MFMA32(tdo[1][0], pb1, dv[kb][0]); M_(6); M_(7); M_(8); M_(9); M_(10); M_(11); tq[1][0] = TRF(qi_, 0, 1); PIN(); \
    dv[kb][1] = MFMA32(tdo[1][1], pb1, dv[kb][1]); M_(12); M_(13); M_(14); M_(15); tq[1][1] = TRF(qi_, 1, 1);    \
    const bf16x8 ds0 = acc_frag(dp, 0);                                                                          \
    PIN();                                                                                                       \
    /* G5 */                                                                                                     \
    const int ex_ = (CUR) * 32768 + (kb) * 2048;                                                                 \
    dk[kb][0] = MFMA32(tq[0][0], ds0, dk[kb][0]);                                                                \
    const bf16x8 ds1 = acc_frag(dp, 1);                                                                          \
    const u32x4_t w0 = __builtin_bit_cast(u32x4_t, ds0), w1 = __builtin_bit_cast(u32x4_t, ds1);                 \
    lds_wr64(exo[0] + ex_, w0[0], w0[1]); lds_wr64(exo[1] + ex_, w0[2], w0[3]); PIN();                           \
    dk[kb][1] = MFMA32(tq[0][1], ds0, dk[kb][1]);                                                                \
    lds_wr64(exo[2] + ex_, w1[0], w1[1]); lds_wr64(exo[3] + ex_, w1[2], w1[3]); PIN();                           \
    dk[kb][0] = MFMA32(tq[1][0], ds1, dk[kb][0]);                                                                \
    if ((kb) < 3) { RD_FD(CUR); } PIN();                                                                         \
    dk[kb][1] = MFMA32(tq[1][1], ds1, dk[kb][1]);                                                                \
    if ((kb) < 2) { _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fk[ks] = lds_rd128(kro[ks] + ((kb) + 2) * 4096); } \
    PIN();                                                                                                       \
    s = sn;                                                                                                      \
  }
  // dQ fragment sets (three, rotating): A = K^T[16 d][32 keys], B0 / B1 = dS^T[32 keys][16 q] of the two query halves
#define DQ_LOAD(SET, PREV, ks_)                                                                                  \
  do {                                                                                                           \
    qa[SET] = lds_tr2(dqa + (ks_) * 4096, dqa + (ks_) * 4096 + 512);                                             \
    qb0[SET] = lds_tr2(dqb[0][0] + (PREV) * 32768 + (ks_) * 2048, dqb[0][1] + (PREV) * 32768 + (ks_) * 2048);    \
    qb1[SET] = lds_tr2(dqb[1][0] + (PREV) * 32768 + (ks_) * 2048, dqb[1][1] + (PREV) * 32768 + (ks_) * 2048);    \
  } while (0)
#define DQ_AHEAD 5   // key steps of transposed reads in flight (nothing else is live in this phase: 12 registers each)
#define DQ_STEP(SET, PREV, ks_)                                                                                  \
  do {                                                                                                           \
    a0 = MFMA16(qa[SET], qb0[SET], a0); a1 = MFMA16(qa[SET], qb1[SET], a1);                                      \
    if ((ks_) + DQ_AHEAD < 16) DQ_LOAD(SET, PREV, (ks_) + DQ_AHEAD);                                             \
    PIN();                                                                                                       \
  } while (0)
#define DQ_PHASE(PREV)                                                                                           \
  do {                                                                                                           \
    bf16x8 qa[DQ_AHEAD], qb0[DQ_AHEAD], qb1[DQ_AHEAD];                                                           \
    _Pragma("unroll") for (int k = 0; k < DQ_AHEAD; ++k) DQ_LOAD(k, PREV, k);                                    \
    PIN();                                                                                                       \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) DQ_STEP(k % DQ_AHEAD, PREV, k);                               \
  } while (0)
  // dQ of the previous block finished: scale, round, store (4 consecutive d of one query per lane)
#define DQ_STORE(qb_)                                                                                            \
  do {                                                                                                           \
    _Pragma("unroll") for (int qg = 0; qg < 2; ++qg) {                                                           \
      const f32x4v a = qg ? a1 : a0;                                                                             \
      const int q_ = (qb_) * 32 + 16 * qg + (lane & 15);                                                         \
      uint2 o_;                                                                                                  \
      o_.x = pack_bf2(a[0] * p.scale, a[1] * p.scale);                                                           \
      o_.y = pack_bf2(a[2] * p.scale, a[3] * p.scale);                                                           \
      if (q_ < S) {                                                                                              \
        const float f0_ = bf_lo(o_.x), f1_ = bf_hi(o_.x), f2_ = bf_lo(o_.y), f3_ = bf_hi(o_.y);                  \
        if (B16) *(uint2*)(g_dq + (uint32_t)(q_ * p.lddqkv + 4 * (lane >> 4))) = o_;                            \
        if (F8) {                                                                                                \
          *(uint32_t*)(g_dq8 + (uint32_t)(q_ * p.lddqkv8 + 4 * (lane >> 4))) =                                   \
              pack_fp8x4(f0_ * qs8, f1_ * qs8, f2_ * qs8, f3_ * qs8, true);                                      \
          amax8 = fmaxf(amax8, fmaxf(fmaxf(fabsf(f0_), fabsf(f1_)), fmaxf(fabsf(f2_), fabsf(f3_))));             \
        }                                                                                                        \
        colq[0] += f0_; colq[1] += f1_; colq[2] += f2_; colq[3] += f3_;                                          \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
  // ---- the query-block loop: iteration i stages block i+1, computes the key-owner products of block i (exchange image
  // i&1) and dQ of block i-1 (image (i-1)&1); one barrier per iteration.
  //  RAW: the stage and the statistics of block i+1 are written during iteration i (DMA + vmcnt(0), ds_write) and read
  //       after its closing barrier; the exchange image written in iteration i is read in iteration i+1.
  //  WAR: the stage / statistics slot re-filled in iteration i and the exchange image re-written in iteration i were last
  //       read in iteration i-1, behind that iteration's closing barrier (which retires the LDS reads: lgkmcnt(0)).
#if FUSED_DBG & 1024   // diagnostic build: shader-clock stamps around the phases of an iteration, summed per wave
  unsigned long long tk_[6] = {0, 0, 0, 0, 0, 0}, tl_ = 0;
#define STAMP(k) do { PIN(); const unsigned long long t_ = __builtin_readcyclecounter(); tk_[k] += t_ - tl_; tl_ = t_; PIN(); } while (0)
#else
#define STAMP(k)
#endif
#define F_ITER(CUR, i_)                                                                                          \
  do {                                                                                                           \
    const int it_ = (i_);                                                                                        \
    STAMP(5);                                                                                                    \
    if (it_ + 1 < NQ) F_STAGE((CUR) ^ 1, it_ + 1);                                                               \
    STAMP(0);                                                                                                    \
    if (wave_has_keys && it_ < NQ) {                                                                             \
      bf16x8 fq[4], fk[4], fd[4];                                                                                \
      f32x4 n0, n1, n2, n3, b0, b1, b2, b3;  /* -delta and the bias of the 16 queries this lane's registers hold */ \
      f32x16 s = zero16();                                                                                       \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                         \
        fq[ks] = lds_rd128(rowo[ks] + (CUR) * 8192); fk[ks] = lds_rd128(kro[ks]);                                \
      }                                                                                                          \
      RD_FD(CUR);                                                                                                \
      PIN();                                                                                                     \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) s = MFMA32(fq[ks], fk[ks], s);   /* S' of key block 0 */  \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fk[ks] = lds_rd128(kro[ks] + 4096);                       \
      PIN();                                                                                                     \
      KO_KB(CUR, 0) KO_KB(CUR, 1) KO_KB(CUR, 2) KO_KB(CUR, 3)                                                    \
    } else {                                                                                                     \
      if (it_ < NQ) { /* the dQ phase of the next iteration reads these rows of the exchange image */            \
        _Pragma("unroll") for (int kb = 0; kb < 4; ++kb)                                                         \
          _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) lds_wr64(exo[g4] + (CUR) * 32768 + kb * 2048, 0u, 0u); \
      }                                                                                                          \
    }                                                                                                            \
    STAMP(1);                                                                                                    \
    if (it_ + 1 < NQ) { F_STAT_LOAD(it_ + 1); PIN(); }  /* the next block's rows of dO and O: the dQ phase covers the latency */ \
    if (it_ >= 1) {                                                                                              \
      f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};                                               \
      DQ_PHASE((CUR) ^ 1);                                                                                       \
      DQ_STORE(it_ - 1);                                                                                         \
    }                                                                                                            \
    STAMP(2);                                                                                                    \
    if (it_ + 1 < NQ) F_STAT_STORE((CUR) ^ 1, it_ + 1);                                                          \
    STAMP(3);                                                                                                    \
    DMA_WAIT();                                                                                                  \
    __syncthreads();                                                                                             \
    STAMP(4);                                                                                                    \
  } while (0)
#if FUSED_DBG & 1024
  const unsigned long long tstart_ = __builtin_readcyclecounter();
  tl_ = tstart_;
#endif
  for (int i = 0; i <= NQ; i += 2) {
    F_ITER(0, i);
    if (i + 1 <= NQ) F_ITER(1, i + 1);
  }
#if FUSED_DBG & 1024
  // stamps leave through p.delta (not otherwise written by this kernel): [block][wave][8] kilo-cycles
  if (lane == 0 && p.delta) {
    float* o_ = p.delta + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int k = 0; k < 6; ++k) o_[k] = (float)tk_[k] * 1e-3f;
    o_[6] = (float)(tl_ - tstart_) * 1e-3f;
  }
#endif

  // ---- epilogue. Query rows past the last block with work: dQ = 0 (the dX GEMM reads every row). Lane-derived values are
  // formed afresh (LANE_ID): nothing of the prologue's lane arithmetic stays live across the loop for these lines.
  LANE_ID(lane_e);
  const int tid_e = wave * 64 + lane_e;
  for (int r = NQ * 32 + (tid_e >> 3); r < S; r += 32) {
    if (B16) *(uint4*)(p.dqkv + (tok0 + r) * p.lddqkv + hd * 64 + (tid_e & 7) * 8) = make_uint4(0, 0, 0, 0);
    if (F8) *(uint2*)(p.dqkv8 + (tok0 + r) * p.lddqkv8 + hd * 64 + (tid_e & 7) * 8) = make_uint2(0, 0);
  }
  // dK, dV of this wave's four key blocks through the per-wave transpose patch (the exchange images are free now)
  bf16_t* patch = (bf16_t*)(smem + FB_EXCH) + wave * (32 * 72);
  const bool accq = p.colpart_accumulate != 0;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int key0 = wave * 128 + kb * 32;
    int rows_valid = S - key0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
    float* cp = (p.colpart && wave < QT) ? p.colpart + ((size_t)(b * QT + wave) * 4 + kb) * (3 * H) + hd * 64 : nullptr;
    if (rows_valid > 0) {
      bf16_t* out = B16 ? p.dqkv + (tok0 + key0) * p.lddqkv + hd * 64 : nullptr;
      uint8_t* out8 = F8 ? p.dqkv8 + (tok0 + key0) * p.lddqkv8 + hd * 64 : nullptr;
      const Out8 ok8 = {out8 ? out8 + H : nullptr, p.lddqkv8, qs8, true}, ov8 = {out8 ? out8 + 2 * H : nullptr, p.lddqkv8, qs8, true};
      store_transposed<false, true>(dk[kb][0], dk[kb][1], p.scale, patch, out ? out + H : nullptr, p.lddqkv, rows_valid, lane_e,
                                    cp ? cp + H : nullptr, accq, F8 ? &ok8 : nullptr, &amax8);
      __builtin_amdgcn_wave_barrier();
      store_transposed<false, true>(dv[kb][0], dv[kb][1], 1.0f, patch, out ? out + 2 * H : nullptr, p.lddqkv, rows_valid, lane_e,
                                    cp ? cp + 2 * H : nullptr, accq, F8 ? &ov8 : nullptr, &amax8);
      __builtin_amdgcn_wave_barrier();
    } else if (cp && !accq) {
      cp[H + lane_e] = 0.f;
      cp[2 * H + lane_e] = 0.f;
    }
  }
  if (F8 && p.dqkv_amax) {
    amax8 = wave_max(amax8);
    if (lane_e == 0) atomic_max_abs(p.dqkv_amax, amax8, blockIdx.x * 4 + wave);
  }
  // bias-gradient partial of the Q block: the sums of this workgroup go to row 0 of the sample's rows, the others get 0
  if (p.colpart) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = colq[r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      colq[r] = v;
    }
    float* row0 = p.colpart + (size_t)(b * QT) * 4 * (3 * H) + hd * 64;
    if ((lane_e & 15) == 0) {
      float* d = row0 + 16 * wave + 4 * (lane_e >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] = accq ? d[r] + colq[r] : colq[r];
    }
    if (!accq)
      for (int rr = 1 + wave; rr < 4 * QT; rr += 4) row0[(size_t)rr * (3 * H) + lane_e] = 0.f;
  }
}

}  // namespace

// S <= 512: the single-kernel form. p->delta is not written (the kernel keeps delta in LDS).
extern "C" int plb_launch_attn_bwd_fused(const PlbAttn* p, hipStream_t stream) {
  if (p->H != p->NH * 64 || p->S < 1 || p->S > 512 || p->B < 1) return 1;
  if (p->ldqkv % 8 || p->ldctx % 8 || p->lddctx % 8 || p->lddqkv % 8) return 1;
  if ((!p->dqkv && !p->dqkv8) || (p->dqkv8 && (!p->dqkv_scale || p->lddqkv8 % 8))) return 1;
  dim3 grid(p->NH * p->B), block(256);
  const double unit = (double)p->B * p->NH * (double)p->S * p->S * 64.0;
  const double io = 2.0 * p->B * p->S * (double)p->H;
  // algorithmic work of the backward = 4 products (dP, dQ, dV, dK); the S recomputation is not credited
  const int tok = plb_prof_begin(PLB_K_ATTN_BWD, stream, 8.0 * unit, 6.0 * io);
  // (bf16 rows AND the image in one call — an fp8 call whose weight gradients read bf16 operands — would need 4 more
  // registers than the file has: that combination takes the two-kernel form, plb_launch_attn_bwd)
  if (p->dqkv && p->dqkv8) return 1;
  if (!p->dqkv8) hipLaunchKernelGGL(attn_bwd_fused_kernel<0>, grid, block, 0, stream, *p);
  else hipLaunchKernelGGL(attn_bwd_fused_kernel<1>, grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
