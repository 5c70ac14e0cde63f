// Fused multi-head attention for the shared ALBERT layer, head_dim 64, gfx950.
// Restates AlbertAttention's softmax(q·k^T·d^-0.5 + key_padding_mask)·v (modeling_albert.py:110-135,
// 166-200) as a flash-style single pass: scores never leave registers.
//
// Forward, one workgroup = 4 waves = 128 query rows of one (batch, head); a wave owns 32 queries.
//   S^T[key][q] = K·Q^T  (MFMA 32x32x16, K rows from LDS, Q fragments in registers) puts one query
//   per lane, so the row max / row sum are lane-local (+ one exchange with lane^32).  The S^T
//   accumulator converted to bf16 is directly the B operand of O^T[dv][q] = V^T·P^T (accumulator-as-
//   operand, k order 16s+8(j>>2)+4h+(j&3)); V^T fragments come from ds_read_b64_tr_b16 on a
//   [4 keys][32 dv] sub-tiled LDS image (each half-wave read = one 256-B bank row).
// Backward, two kernels that recompute P from the saved log-sum-exp:
//   dq kernel  (same tiling as forward): dS^T = P^T∘(dP^T - delta), dQ^T = K^T·dS^T ; also writes delta.
//   dkv kernel (a wave owns 32 keys, loops over query tiles): dV^T = dO^T·P, dK^T = Q^T·dS.
// Key padding comes from lengths[b]; whole key tiles past the length are skipped.
#include "attn_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

// ---------------------------------------------------------------------------------------- forward
#ifndef RESCALE_TH
#define RESCALE_TH 8.0f  // the forward rescales its accumulators when a row maximum grows by more than this (log2 units)
#endif
#ifndef FWD_WAVES
#define FWD_WAVES 3   // waves per SIMD the register allocation is held to
#endif
__global__ __launch_bounds__(256, FWD_WAVES) void attn_fwd_kernel(PlbAttn p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][64 * 64];  // [stage][K row | V tr] 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DBG_EARLY_EXIT(p);
  // 1-D grid, XCD-aware: the q-tiles of one (batch, head) read the same K/V, so they get consecutive
  // logical ids = the same XCD's L2 (block id % 8 labels the XCD; placement affects speed only)
  const int QT = (p.S + 127) >> 7;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = logical % QT, bh = logical / QT;
  const int hd = bh % p.NH, b = bh / p.NH;
  const int S = p.S, H = p.H;
  int len = p.lengths ? p.lengths[b] : S;
  len = len < 1 ? 1 : (len > S ? S : len);
  const int q0 = bx * 128 + wave * 32;
  const size_t tok0 = (size_t)b * S;
  const bf16_t* kbase = p.qkv + H + hd * 64;
  const bf16_t* vbase = p.qkv + 2 * H + hd * 64;
  const int ld = p.ldqkv;
  const int lq = lane & 31, h = lane >> 5;
  // compact-query mode (PlbAttn.qoff): this sample's queries are rows [qlo, qlo + Sq) of p.q; outputs by compact row
  const bool cq = p.qoff != nullptr;
  const int qlo = cq ? p.qoff[b] : 0;
  const int Sq = cq ? p.qoff[b + 1] - qlo : S;
  if (bx * 128 >= Sq) return;   // (wave-uniform, before any barrier: a q tile without a query)
  const size_t qrow0 = cq ? (size_t)qlo : tok0;   // first query row of the sample in q / ctx
  const int ldq = cq ? p.ldq : ld;

  bf16x8 qf[4];
  {
    int qr = q0 + lq; qr = qr < Sq ? qr : Sq - 1;
    const bf16_t* qp = (cq ? p.q : p.qkv) + hd * 64 + (qrow0 + qr) * ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  const int nkt = (len + 63) >> 6;
  const StageLane g = stage_lane(wave, lane);
  const uint32_t vA0 = (uint32_t)(g.rA * ld + g.cA0) * 2, vA1 = (uint32_t)((g.rA + 8) * ld + g.cA1) * 2;
  const uint32_t vT = (uint32_t)(g.rT * ld + g.cT) * 2;
  const uint32_t lds0 = LDS_ADDR(&smem[0][0][0]) + (uint32_t)wave * 2048;
  const bf16_t* gk = kbase + tok0 * ld;
  const bf16_t* gv = vbase + tok0 * ld;
#define KV_STAGE(ST, kt_)                                                                                \
  do {                                                                                                   \
    const int k0_ = (kt_) * 64;                                                                          \
    const uint32_t l_ = lds0 + (ST) * 16384;                                                             \
    if (k0_ + 64 <= S) {                                                                                 \
      const char* sk_ = (const char*)(gk + (size_t)k0_ * ld);                                            \
      const char* sv_ = (const char*)(gv + (size_t)k0_ * ld);                                            \
      DMA16(sk_, vA0, l_); DMA16(sk_, vA1, l_ + 1024);                                                   \
      DMA16(sv_, vT, l_ + 8192); DMA16(sv_ + 16 * ld, vT, l_ + 8192 + 1024);                             \
    } else { /* the tile that crosses S: rows are clamped, every lane computes its own offsets */        \
      const int a0_ = min(k0_ + g.rA, S - 1), a1_ = min(k0_ + g.rA + 8, S - 1);                          \
      const int t0_ = min(k0_ + g.rT, S - 1), t1_ = min(k0_ + g.rT + 8, S - 1);                          \
      DMA16((const char*)gk, (uint32_t)(a0_ * ld + g.cA0) * 2, l_);                                      \
      DMA16((const char*)gk, (uint32_t)(a1_ * ld + g.cA1) * 2, l_ + 1024);                               \
      DMA16((const char*)gv, (uint32_t)(t0_ * ld + g.cT) * 2, l_ + 8192);                                \
      DMA16((const char*)gv, (uint32_t)(t1_ * ld + g.cT) * 2, l_ + 8192 + 1024);                         \
    }                                                                                                    \
  } while (0)

  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = p.scale * LOG2E;

  int kqo[4];
  row_frag_offsets(lane, kqo);
  KV_STAGE(0, 0);
  DMA_WAIT();
  __syncthreads();
  // One tile of 64 keys out of LDS stage CUR (a literal: every LDS address below is a lane constant + an immediate; with
  // a run-time stage hipcc re-derived ~45 address VALU per tile in a VALU-bound loop). The loop runs two tiles per trip.
#define FWD_TILE(CUR, kt)                                                                                         \
  do {                                                                                                            \
    if ((kt) + 1 < nkt) KV_STAGE((CUR) ^ 1, (kt) + 1);                                                            \
    const bf16_t* sK = smem[CUR][0];                                                                              \
    const bf16_t* sV = smem[CUR][1];                                                                              \
    f32x16 s0 = zero16(), s1 = zero16();                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                            \
      s0 = MFMA32(ROW_FRAG(sK, 0, ks, kqo), qf[ks], s0);                \
      s1 = MFMA32(ROW_FRAG(sK, 1, ks, kqo), qf[ks], s1);                \
    }                                                                                                             \
    /* Softmax in the exp2 domain. The loop is VALU-bound at head_dim 64 (one v_exp per score against 256 MFMA */ \
    /* flops), so: masking only in a tile that crosses the length, the scale folded into one FMA per score (max */\
    /* taken on raw scores: the scale is positive), the row maximum as a chain of three-input maxima, and the */   \
    /* accumulator rescaled only when some query's running max has moved by more than RESCALE_TH (log2 units): below */ \
    /* that the stale maximum stays the reference — probabilities up to 2^TH instead of 1, harmless in fp32 / bf16, */ \
    /* and LSE = m_run + log2(l) holds for any reference. Both conditionals carry an empty asm statement: without it */ \
    /* hipcc if-converts them, and the 97 compare / select / index VALU of the mask and the 16 packed multiplies of */  \
    /* the rescale then run in EVERY tile of a loop whose SIMDs are VALU-busy 61 % of the time (SQ_ACTIVE_INST_VALU). */ \
    if ((kt) * 64 + 64 > len) {                                                                                   \
      asm volatile("; tile crosses the length");                                                                  \
      const int kbase_i = (kt) * 64 + 4 * h;                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
        const int kr0 = kbase_i + (r & 3) + 8 * (r >> 2);                                                         \
        s0[r] = (kr0 < len) ? s0[r] : -INFINITY;                                                                  \
        s1[r] = (kr0 + 32 < len) ? s1[r] : -INFINITY;                                                             \
      }                                                                                                           \
    }                                                                                                             \
    float mx = fmaxf(s0[0], s1[0]);                                                                               \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s0[r]), s1[r]);       \
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                                                                       \
    const float m_new = fmaxf(m_run, mx * sl2); /* finite: the first tile always holds key 0 < len */             \
    if (__builtin_amdgcn_ballot_w64(m_new > m_run + RESCALE_TH) != 0) { /* first tile: m_run = -inf */            \
      asm volatile("; rescale");                                                                                  \
      const float alpha = EXP2(m_run - m_new);                                                  \
      l_run *= alpha;                                                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }                          \
      m_run = m_new;                                                                                              \
    }                                                                                                             \
    f32x2_t ls2 = {0.f, 0.f}; /* row sums as packed adds: one v_pk_add_f32 per pair of scores */                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                              \
      s0[r] = EXP2(__builtin_fmaf(s0[r], sl2, -m_run));                                         \
      s1[r] = EXP2(__builtin_fmaf(s1[r], sl2, -m_run));                                         \
      ls2 += f32x2_t{s0[r], s1[r]};                                                                               \
    }                                                                                                             \
    l_run += ls2[0] + ls2[1];                                                                                     \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                            \
      const bf16x8 pb = acc_frag((s4 >> 1) ? s1 : s0, s4 & 1);                                                    \
      o0 = MFMA32(tr_frag(sV, s4 >> 1, s4 & 1, 0, lane), pb, o0);       \
      o1 = MFMA32(tr_frag(sV, s4 >> 1, s4 & 1, 1, lane), pb, o1);       \
    }                                                                                                             \
    DMA_WAIT();                                                                                                   \
    TILE_SYNC();                                                                                                 \
  } while (0)
  for (int kt = 0; kt < DBG_TILES(nkt); kt += 2) {
    FWD_TILE(0, kt);
    if (kt + 1 < nkt) FWD_TILE(1, kt + 1);
  }
#undef FWD_TILE
#undef KV_STAGE
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (h == 0 && q0 + lq < Sq)
    p.lse[(cq ? (size_t)hd * p.nq_total + qlo : ((size_t)b * p.NH + hd) * S) + q0 + lq] = -(m_run + __log2f(l_tot)) / sl2;  // see PlbAttn.lse
  // all waves are past the last barrier: reuse the staging LDS as per-wave transpose patches
  bf16_t* patch = &smem[0][0][0] + wave * (32 * 72);
  int rows_valid = Sq - q0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
  // fp8 mode: the e4m3 image of the context rows for the fp8 dense projection and its weight gradient
  const Out8 o8 = {p.ctx8 ? p.ctx8 + (qrow0 + q0) * p.ldctx8 + hd * 64 : nullptr, p.ldctx8, p.ctx8 ? p.ctx_scale[0] : 1.0f, false};
  float amax8 = 0.f;
  if (rows_valid > 0)
    store_transposed<(ATTN_OUT_NT & 1) != 0>(o0, o1, inv, patch, p.ctx + (qrow0 + q0) * p.ldctx + hd * 64, p.ldctx, rows_valid, lane,
                                             nullptr, false, &o8, &amax8);
  if (p.ctx8 && p.ctx_amax) {
    amax8 = wave_max(amax8);
    if (lane == 0) atomic_max_abs(p.ctx_amax, amax8, blockIdx.x * 4 + wave);
  }
}

// ------------------------------------------------------------------------------------- backward dQ
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(PlbAttn p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][3][64 * 64];  // [stage][K row | K tr | V row] 48 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DBG_EARLY_EXIT(p);
  // 1-D grid, XCD-aware: the q-tiles of one (batch, head) read the same K/V, so they get consecutive
  // logical ids = the same XCD's L2 (block id % 8 labels the XCD; placement affects speed only)
  const int QT = (p.S + 127) >> 7;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = logical % QT, bh = logical / QT;
  const int hd = bh % p.NH, b = bh / p.NH;
  const int S = p.S, H = p.H;
  int len = p.lengths ? p.lengths[b] : S;
  len = len < 1 ? 1 : (len > S ? S : len);
  const int q0 = bx * 128 + wave * 32;
  const size_t tok0 = (size_t)b * S;
  const bf16_t* kbase = p.qkv + H + hd * 64;
  const bf16_t* vbase = p.qkv + 2 * H + hd * 64;
  const int ld = p.ldqkv;
  const int lq = lane & 31, h = lane >> 5;
  // compact-query mode (PlbAttn.qoff): see the forward
  const bool cq = p.qoff != nullptr;
  const int qlo = cq ? p.qoff[b] : 0;
  const int Sq = cq ? p.qoff[b + 1] - qlo : S;
  if (bx * 128 >= Sq) {   // a q tile without a query (compact mode only): its bias-gradient partial rows are zeros
    if (p.colpart && !p.colpart_accumulate) p.colpart[((size_t)(b * QT + bx) * 4 + wave) * (3 * H) + hd * 64 + lane] = 0.f;
    return;
  }
  const size_t qrow0 = cq ? (size_t)qlo : tok0;
  const int ldq = cq ? p.ldq : ld;
  int qr = q0 + lq; qr = qr < Sq ? qr : Sq - 1;

  bf16x8 qf[4], dof[4];
  float delta;
  {
    const bf16_t* qp = (cq ? p.q : p.qkv) + hd * 64 + (qrow0 + qr) * ldq + 8 * h;
    const bf16_t* dop = p.dctx + (qrow0 + qr) * p.lddctx + hd * 64 + 8 * h;
    const bf16_t* op = p.ctx + (qrow0 + qr) * p.ldctx + hd * 64 + 8 * h;
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const bf16x8*)(qp + ks * 16);
      dof[ks] = *(const bf16x8*)(dop + ks * 16);
      bf16x8 of = *(const bf16x8*)(op + ks * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) d += bf2f((bf16_t)dof[ks][j]) * bf2f((bf16_t)of[j]);
    }
    delta = d + __shfl_xor(d, 32, 64);
  }
  const size_t stat = (cq ? (size_t)hd * p.nq_total + qlo : ((size_t)b * p.NH + hd) * S) + qr;
  if (h == 0 && q0 + lq < Sq) p.delta[stat] = -delta;   // stored NEGATED: the dK/dV kernel starts its dP accumulators there
  const float lse2 = p.lse[stat] * (p.scale * LOG2E);  // stored as -LSE in raw-score units (PlbAttn.lse)
  const float sl2 = p.scale * LOG2E;

  const int nkt = (len + 63) >> 6;
  const StageLane g = stage_lane(wave, lane);
  const uint32_t vA0 = (uint32_t)(g.rA * ld + g.cA0) * 2, vA1 = (uint32_t)((g.rA + 8) * ld + g.cA1) * 2;
  const uint32_t vT = (uint32_t)(g.rT * ld + g.cT) * 2;
  const uint32_t lds0 = LDS_ADDR(&smem[0][0][0]) + (uint32_t)wave * 2048;
  const bf16_t* gk = kbase + tok0 * ld;
  const bf16_t* gv = vbase + tok0 * ld;
  // [K row | K tr | V row]: the V row image takes the K row image's lane offsets from V's base
#define KV_STAGE(ST, kt_)                                                                                \
  do {                                                                                                   \
    const int k0_ = (kt_) * 64;                                                                          \
    const uint32_t l_ = lds0 + (ST) * 24576;                                                             \
    if (k0_ + 64 <= S) {                                                                                 \
      const char* sk_ = (const char*)(gk + (size_t)k0_ * ld);                                            \
      const char* sv_ = (const char*)(gv + (size_t)k0_ * ld);                                            \
      DMA16(sk_, vA0, l_); DMA16(sk_, vA1, l_ + 1024);                                                   \
      DMA16(sk_, vT, l_ + 8192); DMA16(sk_ + 16 * ld, vT, l_ + 8192 + 1024);                             \
      DMA16(sv_, vA0, l_ + 16384); DMA16(sv_, vA1, l_ + 16384 + 1024);                                   \
    } else { /* the tile that crosses S: rows are clamped, every lane computes its own offsets */        \
      const int a0_ = min(k0_ + g.rA, S - 1), a1_ = min(k0_ + g.rA + 8, S - 1);                          \
      const int t0_ = min(k0_ + g.rT, S - 1), t1_ = min(k0_ + g.rT + 8, S - 1);                          \
      const uint32_t va0_ = (uint32_t)(a0_ * ld + g.cA0) * 2, va1_ = (uint32_t)(a1_ * ld + g.cA1) * 2;  \
      DMA16((const char*)gk, va0_, l_); DMA16((const char*)gk, va1_, l_ + 1024);                         \
      DMA16((const char*)gk, (uint32_t)(t0_ * ld + g.cT) * 2, l_ + 8192);                                \
      DMA16((const char*)gk, (uint32_t)(t1_ * ld + g.cT) * 2, l_ + 8192 + 1024);                         \
      DMA16((const char*)gv, va0_, l_ + 16384); DMA16((const char*)gv, va1_, l_ + 16384 + 1024);         \
    }                                                                                                    \
  } while (0)

  f32x16 dq0 = zero16(), dq1 = zero16();
  int kro[4];
  row_frag_offsets(lane, kro);
  KV_STAGE(0, 0);
  DMA_WAIT();
  __syncthreads();
  // One tile of 64 keys out of LDS stage CUR (a literal, as in the forward: every LDS address is a lane constant + an
  // immediate). Row constants as the initial accumulator: the query sits on the lane, so dP starts at -delta and the MFMA
  // chain leaves dP - delta ready (one VALU less per score).
#define DQ_TILE(CUR, kt_)                                                                                 \
  do {                                                                                                    \
    const int kt = (kt_);                                                                                 \
    if (kt + 1 < nkt) KV_STAGE((CUR) ^ 1, kt + 1);                                                        \
    const bf16_t* sK = smem[CUR][0];                                                                      \
    const bf16_t* sKt = smem[CUR][1];                                                                     \
    const bf16_t* sV = smem[CUR][2];                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                    \
      f32x16 s = zero16(), dp = splat16(-delta);                                                          \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                  \
        s = MFMA32(ROW_FRAG(sK, kb, ks, kro), qf[ks], s);                                                 \
        dp = MFMA32(ROW_FRAG(sV, kb, ks, kro), dof[ks], dp);                                              \
      }                                                                                                   \
      if (kt * 64 + 64 > len) {                                                                           \
        asm volatile("; tile crosses the length"); /* keeps the branch: see the forward */                \
        const int kb0 = kt * 64 + kb * 32 + 4 * h;                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
          const int key = kb0 + (r & 3) + 8 * (r >> 2);                                                   \
          s[r] = (key < len) ? s[r] : -INFINITY; /* exp2(-inf) = 0 */                                     \
        }                                                                                                 \
      }                                                                                                   \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
        const float pr = EXP2(__builtin_fmaf(s[r], sl2, lse2));                                           \
        s[r] = pr * dp[r]; /* dS^T = P (dP - delta); the scale is applied once at the end */              \
      }                                                                                                   \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                  \
        const bf16x8 dsb = acc_frag(s, s2);                                                               \
        dq0 = MFMA32(tr_frag(sKt, kb, s2, 0, lane), dsb, dq0);                                            \
        dq1 = MFMA32(tr_frag(sKt, kb, s2, 1, lane), dsb, dq1);                                            \
      }                                                                                                   \
    }                                                                                                     \
    DMA_WAIT();                                                                                           \
    TILE_SYNC();                                                                                          \
  } while (0)
  for (int kt2 = 0; kt2 < DBG_TILES(nkt); kt2 += 2) {
    DQ_TILE(0, kt2);
    if (kt2 + 1 < nkt) DQ_TILE(1, kt2 + 1);
  }
#undef DQ_TILE
#undef KV_STAGE
  bf16_t* patch = &smem[0][0][0] + wave * (32 * 72);
  int rows_valid = Sq - q0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
  // bias-gradient partial row of this wave: [(b * QT + q tile) * 4 + wave][3H], columns hd*64.. of the Q block
  float* cp = p.colpart ? p.colpart + ((size_t)(b * QT + bx) * 4 + wave) * (3 * H) + hd * 64 : nullptr;
  // fp8 mode: the e5m2 image of the gradient rows for the fp8 dX GEMM and the QKV weight gradient (dqkv itself may be off).
  // Compact-query mode: dQ rows (and their image) go to the compact buffers, not into dqkv's Q block
  bf16_t* const dq_out = cq ? (p.dq ? p.dq + (qrow0 + q0) * p.lddq + hd * 64 : nullptr)
                            : (p.dqkv ? p.dqkv + (tok0 + q0) * p.lddqkv + hd * 64 : nullptr);
  uint8_t* const dq8_out = cq ? (p.dq8 ? p.dq8 + (qrow0 + q0) * p.lddq8 + hd * 64 : nullptr)
                              : (p.dqkv8 ? p.dqkv8 + (tok0 + q0) * p.lddqkv8 + hd * 64 : nullptr);
  const Out8 o8 = {dq8_out, cq ? p.lddq8 : p.lddqkv8, dq8_out ? p.dqkv_scale[0] : 1.0f, true};
  float amax8 = 0.f;
  if (rows_valid > 0)
    store_transposed<(ATTN_OUT_NT & 2) != 0, true>(dq0, dq1, p.scale, patch, dq_out, cq ? p.lddq : p.lddqkv,
                     rows_valid, lane, cp, p.colpart_accumulate != 0, &o8, &amax8);
  else if (cp && !p.colpart_accumulate) cp[lane] = 0.f;
  if (dq8_out && p.dqkv_amax) {
    amax8 = wave_max(amax8);
    if (lane == 0) atomic_max_abs(p.dqkv_amax, amax8, blockIdx.x * 4 + wave);
  }
}

// ---------------------------------------------------------------------------------- backward dK,dV
// A wave owns 32 keys (K, V fragments in registers, dK^T / dV^T accumulators) and walks the query tiles of its (batch,
// head). A query tile = four 8-KiB LDS images (Q and dO, each as a row image for the S / dP products and as a transposed-
// read image for dV^T = dO^T·P and dK^T = Q^T·dS) plus the 64 row statistics (log2-domain LSE, delta), all written by
// LDS-DMA straight from global memory (global_load_lds, 1 KiB per wave instruction, the images' swizzles applied to the
// per-lane SOURCE address): no staging registers, no ds_write, no per-tile address arithmetic beyond one 64-bit add per
// instruction. Two stages: the DMA of tile t+1 is issued when tile t starts and waited for at the barrier that ends it.

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(PlbAttn p) {
  // [stage][Q row | Q tr | dO row | dO tr] + lse/delta rows
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][4][64 * 64];  // 64 KiB
  __shared__ __attribute__((aligned(16))) float sstat[2][2][64];       // [stage][lse (log2 domain) | delta][q]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DBG_EARLY_EXIT(p);
  const int QT = (p.S + 127) >> 7;  // key tiles of one (batch, head) share Q / dO: same XCD
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = logical % QT, bh = logical / QT;
  const int hd = bh % p.NH, b = bh / p.NH;
  const int S = p.S, H = p.H;
  int len = p.lengths ? p.lengths[b] : S;
  len = len < 1 ? 1 : (len > S ? S : len);
  const int key0 = bx * 128 + wave * 32;
  const size_t tok0 = (size_t)b * S;
  const int ld = p.ldqkv, ldo = p.lddctx;
  const int lk = lane & 31, h = lane >> 5;
  const int mykey = key0 + lk;

  bf16x8 kf[4], vf[4];
  {
    int kr_ = mykey < S ? mykey : S - 1;
    const bf16_t* kp = p.qkv + H + hd * 64 + (tok0 + kr_) * ld + 8 * h;
    const bf16_t* vp = p.qkv + 2 * H + hd * 64 + (tok0 + kr_) * ld + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const bf16x8*)(kp + ks * 16);
      vf[ks] = *(const bf16x8*)(vp + ks * 16);
    }
  }
  const bool key_ok = mykey < len;
  const float sl2 = p.scale * LOG2E;

  // compact-query mode (PlbAttn.qoff): the queries are rows [qlo, qlo + qrows) of p.q / dctx, statistics [NH][Nq]
  const bool cq = p.qoff != nullptr;
  const int qlo = cq ? p.qoff[b] : 0;
  const int Sq = cq ? p.qoff[b + 1] - qlo : S;    // query rows that exist (clamp bound of the staging)
  const int qlen = cq ? Sq : len;                 // query rows that count
  const int ldq = cq ? p.ldq : ld;
  // queries past the length carry exactly zero dO in this model (no loss there), so tiles stop at len
  const int nqt = (bx * 128 < len) ? ((qlen + 63) >> 6) : 0;
  const bf16_t* gq = cq ? p.q + hd * 64 + (size_t)qlo * ldq : p.qkv + hd * 64 + tok0 * ld;
  const bf16_t* gdo = p.dctx + hd * 64 + (cq ? (size_t)qlo : tok0) * ldo;
  const float* glse = p.lse + (cq ? (size_t)hd * p.nq_total + qlo : ((size_t)b * p.NH + hd) * S);
  const float* gdelta = p.delta + (cq ? (size_t)hd * p.nq_total + qlo : ((size_t)b * p.NH + hd) * S);
  // staging: this wave writes rows [16w, 16w+16) of every image, 8 rows (1 KiB) per instruction.
  //  row image  (row_off):  LDS (row, chunk') <- source chunk chunk' ^ ((row >> 1) & 7)
  //  tr image   (tr_off):   LDS 16-byte unit u of 256-byte sub-tile t <- source (row 4(t>>1) + (u>>2), col 32(t&1) + 8(u&3))
  const int rA = wave * 16 + (lane >> 3);
  const int cA0 = ((lane & 7) ^ (lane >> 4)) * 8, cA1 = ((lane & 7) ^ (4 + (lane >> 4))) * 8;
  const int rT = wave * 16 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int cT = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  // byte offsets from the tile's first row; the second instruction of an image is 8 rows further (the row image also
  // changes its swizzled column, so it has its own offset; the tr image uses the scalar base + 8 rows)
  const uint32_t vqA0 = (uint32_t)(rA * ldq + cA0) * 2, vqA1 = (uint32_t)((rA + 8) * ldq + cA1) * 2, vqT = (uint32_t)(rT * ldq + cT) * 2;
  const uint32_t vdA0 = (uint32_t)(rA * ldo + cA0) * 2, vdA1 = (uint32_t)((rA + 8) * ldo + cA1) * 2, vdT = (uint32_t)(rT * ldo + cT) * 2;
  const uint32_t vst = (uint32_t)lane * 4;
  const uint32_t lds0 = LDS_ADDR(&smem[0][0][0]) + (uint32_t)wave * 2048, ldst = LDS_ADDR(&sstat[0][0][0]);
#define DKV_STAGE(ST, qt_)                                                                              \
  do {                                                                                                  \
    const int q0_ = (qt_) * 64;                                                                         \
    const uint32_t l_ = lds0 + (ST) * 32768;                                                            \
    if (q0_ + 64 <= Sq) {                                                                               \
      const char* sq_ = (const char*)(gq + (size_t)q0_ * ldq);                                          \
      const char* sd_ = (const char*)(gdo + (size_t)q0_ * ldo);                                         \
      const char* sq8_ = sq_ + 16 * ldq;                                                                \
      const char* sd8_ = sd_ + 16 * ldo;                                                                \
      DMA16(sq_, vqA0, l_); DMA16(sq_, vqA1, l_ + 1024);                                                \
      DMA16(sq_, vqT, l_ + 8192); DMA16(sq8_, vqT, l_ + 8192 + 1024);                                   \
      DMA16(sd_, vdA0, l_ + 16384); DMA16(sd_, vdA1, l_ + 16384 + 1024);                                \
      DMA16(sd_, vdT, l_ + 24576); DMA16(sd8_, vdT, l_ + 24576 + 1024);                                 \
      if (wave == 0) DMA4((const char*)(glse + q0_), vst, ldst + (ST) * 512);                           \
      if (wave == 1) DMA4((const char*)(gdelta + q0_), vst, ldst + (ST) * 512 + 256);                   \
    } else { /* the tile that crosses S: rows are clamped, every lane computes its own offsets */       \
      const int a0_ = min(q0_ + rA, Sq - 1), a1_ = min(q0_ + rA + 8, Sq - 1);                           \
      const int t0_ = min(q0_ + rT, Sq - 1), t1_ = min(q0_ + rT + 8, Sq - 1);                           \
      const char* sq_ = (const char*)gq;                                                                \
      const char* sd_ = (const char*)gdo;                                                               \
      DMA16(sq_, (uint32_t)(a0_ * ldq + cA0) * 2, l_); DMA16(sq_, (uint32_t)(a1_ * ldq + cA1) * 2, l_ + 1024);                \
      DMA16(sq_, (uint32_t)(t0_ * ldq + cT) * 2, l_ + 8192); DMA16(sq_, (uint32_t)(t1_ * ldq + cT) * 2, l_ + 8192 + 1024);    \
      DMA16(sd_, (uint32_t)(a0_ * ldo + cA0) * 2, l_ + 16384); DMA16(sd_, (uint32_t)(a1_ * ldo + cA1) * 2, l_ + 16384 + 1024); \
      DMA16(sd_, (uint32_t)(t0_ * ldo + cT) * 2, l_ + 24576); DMA16(sd_, (uint32_t)(t1_ * ldo + cT) * 2, l_ + 24576 + 1024);   \
      const uint32_t vs_ = (uint32_t)min(q0_ + lane, Sq - 1) * 4;                                       \
      if (wave == 0) DMA4((const char*)glse, vs_, ldst + (ST) * 512);                                   \
      if (wave == 1) DMA4((const char*)gdelta, vs_, ldst + (ST) * 512 + 256);                           \
    }                                                                                                   \
  } while (0)

  int roff[4];
  row_frag_offsets(lane, roff);
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  if (nqt > 0) DKV_STAGE(0, 0);
  DMA_WAIT();
  __syncthreads();
  // One tile out of stage CUR (a literal: every LDS address is a lane constant + an immediate). The tile body exists
  // twice more: tiles that cross the length (of keys or of queries) mask, all the others run without a single compare /
  // select — written as one select on a wave-uniform flag, hipcc if-converted the mask into every tile.
  // Row constants as initial accumulators: dP starts at -delta[q] (q runs over the registers here; the dQ kernel stores
  // delta negated, so the statistics go into the accumulators as they are), so the MFMA chain leaves dP - delta. Per 32-query block: statistics and the 8 row fragments are read, 8 MFMAs (S, dP), then the 8
  // transposed fragments of the second products are requested BEFORE the softmax arithmetic so they land under it.
#define DKV_BLOCK(CUR, MASK, qb)                                                                                \
  {                                                                                                             \
    f32x16 s, dp;                                                                                               \
    _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                                          \
      const int ql = (qb) * 32 + 8 * rg + 4 * h;                                                                \
      const float4 l4 = *(const float4*)&sstat[CUR][0][ql];                                                     \
      const float4 d4 = *(const float4*)&sstat[CUR][1][ql];                                                     \
      s[4 * rg + 0] = l4.x; s[4 * rg + 1] = l4.y; s[4 * rg + 2] = l4.z; s[4 * rg + 3] = l4.w;                   \
      dp[4 * rg + 0] = d4.x; dp[4 * rg + 1] = d4.y; dp[4 * rg + 2] = d4.z; dp[4 * rg + 3] = d4.w;               \
    }                                                                                                           \
    bf16x8 fq[4], fd[4];                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                          \
      fq[ks] = ROW_FRAG(smem[CUR][0], qb, ks, roff);                                                            \
      fd[ks] = ROW_FRAG(smem[CUR][2], qb, ks, roff);                                                            \
    }                                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                          \
      s = MFMA32(fq[ks], kf[ks], s);                                                                            \
      dp = MFMA32(fd[ks], vf[ks], dp);                                                                          \
    }                                                                                                           \
    bf16x8 tdo[2][2], tq[2][2];                                                                                 \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                            \
      _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) tdo[s2][cb] = tr_frag(smem[CUR][3], qb, s2, cb, lane);   \
    ;                                                                                                           \
    /* s[r] = S[q][key] - LSE: key on the lane, q = qt*64 + qb*32 + (r&3) + 8(r>>2) + 4h */                     \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
      float pr = EXP2(s[r] * sl2);                                                                              \
      if (MASK) pr = (key_ok && (qt * 64 + (qb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < qlen)) ? pr : 0.f;      \
      s[r] = pr;                                                                                                \
    }                                                                                                           \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                            \
      _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) tq[s2][cb] = tr_frag(smem[CUR][1], qb, s2, cb, lane);    \
    {                                                                                                           \
      const bf16x8 pb0 = acc_frag(s, 0), pb1 = acc_frag(s, 1);                                                  \
      dv0 = MFMA32(tdo[0][0], pb0, dv0);                                                                        \
      dv1 = MFMA32(tdo[0][1], pb0, dv1);                                                                        \
      dv0 = MFMA32(tdo[1][0], pb1, dv0);                                                                        \
      dv1 = MFMA32(tdo[1][1], pb1, dv1);                                                                        \
    }                                                                                                           \
    /* dS = P (dP - delta) overwrites dP while the dV MFMAs run */                                              \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) dp[r] = s[r] * dp[r];                                        \
    {                                                                                                           \
      const bf16x8 ds0 = acc_frag(dp, 0), ds1 = acc_frag(dp, 1);                                                \
      dk0 = MFMA32(tq[0][0], ds0, dk0);                                                                         \
      dk1 = MFMA32(tq[0][1], ds0, dk1);                                                                         \
      dk0 = MFMA32(tq[1][0], ds1, dk0);                                                                         \
      dk1 = MFMA32(tq[1][1], ds1, dk1);                                                                         \
    }                                                                                                           \
    ;                                                                                                           \
  }
#define DKV_TILE(CUR, qt_)                                                                                      \
  do {                                                                                                          \
    const int qt = (qt_);                                                                                       \
    if (qt + 1 < nqt) DKV_STAGE((CUR) ^ 1, qt + 1);                                                             \
    const bool need_mask = (key0 + 32 > len) || (qt * 64 + 64 > qlen); /* wave-uniform */                       \
    if (need_mask) { DKV_BLOCK(CUR, true, 0) DKV_BLOCK(CUR, true, 1) }                                          \
    else { DKV_BLOCK(CUR, false, 0) DKV_BLOCK(CUR, false, 1) }                                                  \
    DMA_WAIT();                                                                                                 \
    TILE_SYNC();                                                                                                \
  } while (0)
  for (int qt2 = 0; qt2 < DBG_TILES(nqt); qt2 += 2) {
    DKV_TILE(0, qt2);
    if (qt2 + 1 < nqt) DKV_TILE(1, qt2 + 1);
  }
#undef DKV_TILE
#undef DKV_BLOCK
#undef DKV_STAGE
  bf16_t* patch = &smem[0][0][0] + wave * (32 * 72);
  int rows_valid = S - key0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
  float* cp = p.colpart ? p.colpart + ((size_t)(b * QT + bx) * 4 + wave) * (3 * H) + hd * 64 : nullptr;
  // the lane id is re-derived here (mbcnt, behind an opaque copy): carried across the loop for these few lines it — and
  // the thread id it comes from — were the kernel's two spilled registers (256 VGPRs in the loop)
  int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane_e));
  float amax8 = 0.f;
  if (rows_valid > 0) {
    bf16_t* out = p.dqkv ? p.dqkv + (tok0 + key0) * p.lddqkv + hd * 64 : nullptr;
    uint8_t* out8 = p.dqkv8 ? p.dqkv8 + (tok0 + key0) * p.lddqkv8 + hd * 64 : nullptr;
    const float qs8 = p.dqkv8 ? p.dqkv_scale[0] : 1.0f;
    const Out8 ok8 = {out8 ? out8 + H : nullptr, p.lddqkv8, qs8, true}, ov8 = {out8 ? out8 + 2 * H : nullptr, p.lddqkv8, qs8, true};
    const bool accq = p.colpart_accumulate != 0;
    store_transposed<(ATTN_OUT_NT & 4) != 0, true>(dk0, dk1, p.scale, patch, out ? out + H : nullptr, p.lddqkv, rows_valid, lane_e,
                                             cp ? cp + H : nullptr, accq, &ok8, &amax8);
    __builtin_amdgcn_wave_barrier();
    store_transposed<(ATTN_OUT_NT & 4) != 0, true>(dv0, dv1, 1.0f, patch, out ? out + 2 * H : nullptr, p.lddqkv, rows_valid, lane_e,
                                             cp ? cp + 2 * H : nullptr, accq, &ov8, &amax8);
  } else if (cp && !p.colpart_accumulate) {
    cp[H + lane_e] = 0.f;
    cp[2 * H + lane_e] = 0.f;
  }
  if (p.dqkv8 && p.dqkv_amax) {
    amax8 = wave_max(amax8);
    if (lane_e == 0) atomic_max_abs(p.dqkv_amax, amax8, blockIdx.x * 4 + wave);
  }
}

}  // namespace

static int check_attn(const PlbAttn* p) {
  if (p->H != p->NH * 64 || p->S < 1 || p->B < 1) return 1;
  if (p->ldqkv % 8 || p->ldctx % 8) return 1;
  return 0;
}

extern "C" int plb_launch_attn_fwd(const PlbAttn* p, hipStream_t stream) {
  if (check_attn(p) || (p->ctx8 && (!p->ctx_scale || p->ldctx8 % 8))) return 1;
  if (p->qoff && (!p->q || p->ldq % 8 || p->nq_total < 0)) return 1;
  dim3 grid(((p->S + 127) / 128) * p->NH * p->B), block(256);
  const double unit = (double)p->B * p->NH * (double)p->S * p->S * 64.0;
  const double io = 2.0 * p->B * p->S * (double)p->H;
  const int tok = plb_prof_begin(PLB_K_ATTN_FWD, stream, 4.0 * unit, 4.0 * io);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Two forms of the backward. The two-kernel form (dQ kernel + dK/dV kernel, any S) runs 4 workgroups per CU and packs
// any grid; the single-kernel form (attn_bwd_fused.hip, S <= 512: five products instead of seven, one exponential pass)
// needs a whole CU per (batch, head), so it pays only where B x heads fills the 256 CUs in (nearly) whole rounds —
// measured (profiles/r03_attn_bwd_fused_vs_split.txt, r05_attn_bwd_policy_ab.txt): 16 x 16 heads = exactly one round: 74-78
// vs 83 us per launch, -0.35 ms per step at config D (bf16; fp8 -0.23); 32 x 12 = 1.5 rounds: 143-149 vs 127 us.
// Policy per call (PLBERT_ATTN_BWD = fused | split | auto | hybrid, default auto; plb_set_attn_bwd_fused(1 / 0 / -1 / 2)):
//   auto:   S in (384, 512] and the last round of B x heads at least 90 % full -> fused; everything else -> two kernels.
//           An fp8 call that needs bf16 rows AND the image -> two kernels.
//   hybrid: as auto, and a batch with at least one (nearly) full round of whole samples plus a short remainder is split
//           BY SAMPLE: floor(256 / heads) samples per round to the fused kernel, the rest to the two kernels. Built and
//           measured in round 5 at config A (21 + 11 samples): +0.26 ms per step in bf16, +0.30 in fp8 — the remainder's
//           528 short workgroups cost 70 us, not the 44 their share of the full grid suggests. Off; kept for the record.
static int g_bwd_fused = -2;   // -2: read the environment; -1 auto, 0 split, 1 fused, 2 auto + hybrid
extern "C" void plb_set_attn_bwd_fused(int on) { g_bwd_fused = on < 0 ? -1 : (on > 2 ? 1 : on); }
static int launch_attn_bwd_split(const PlbAttn* p, hipStream_t stream) {
  dim3 grid(((p->S + 127) / 128) * p->NH * p->B), block(256);
  // algorithmic work of the backward = 4 products (dP, dQ, dV, dK); the S recomputation in each
  // kernel and the second dP are not credited
  const double unit = (double)p->B * p->NH * (double)p->S * p->S * 64.0;
  const double io = 2.0 * p->B * p->S * (double)p->H;
  int tok = plb_prof_begin(PLB_K_ATTN_BWD_DQ, stream, 4.0 * unit, 6.0 * io);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  tok = plb_prof_begin(PLB_K_ATTN_BWD_DKV, stream, 4.0 * unit, 6.0 * io);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
// samples [b0, b0 + nb) of a call as a call of their own
static PlbAttn attn_samples(const PlbAttn* p, int b0, int nb) {
  PlbAttn q = *p;
  const size_t t0 = (size_t)b0 * p->S;
  q.B = nb;
  q.qkv = p->qkv + t0 * p->ldqkv;
  q.ctx = p->ctx + t0 * p->ldctx;
  q.dctx = p->dctx + t0 * p->lddctx;
  if (p->dqkv) q.dqkv = p->dqkv + t0 * p->lddqkv;
  if (p->dqkv8) q.dqkv8 = p->dqkv8 + t0 * p->lddqkv8;
  if (p->lengths) q.lengths = p->lengths + b0;
  q.lse = p->lse + (size_t)b0 * p->NH * p->S;
  if (p->delta) q.delta = p->delta + (size_t)b0 * p->NH * p->S;
  if (p->colpart) q.colpart = p->colpart + (size_t)b0 * ((p->S + 127) / 128) * 4 * (3 * p->H);
  return q;
}

extern "C" int plb_launch_attn_bwd(const PlbAttn* p, hipStream_t stream) {
  if (check_attn(p) || p->lddctx % 8 || p->lddqkv % 8 || (!p->dqkv && !p->dqkv8) || (p->dqkv8 && (!p->dqkv_scale || p->lddqkv8 % 8))) return 1;
  if (g_bwd_fused == -2) {
    const char* e = getenv("PLBERT_ATTN_BWD");
    g_bwd_fused = (e && !strcmp(e, "fused")) ? 1 : (e && !strcmp(e, "split")) ? 0 : (e && !strcmp(e, "hybrid")) ? 2 : -1;
  }
  const bool hybrid_on = g_bwd_fused == 2;
  if (p->qoff && (!p->q || p->ldq % 8 || p->nq_total < 0 || (!p->dq && !p->dq8) || (p->dq && p->lddq % 8) || (p->dq8 && p->lddq8 % 8))) return 1;
  const bool can_fuse = p->S <= 512 && !(p->dqkv && p->dqkv8) && !p->qoff;   // (compact queries: the two-kernel form only)
  if (!can_fuse || g_bwd_fused == 0) return launch_attn_bwd_split(p, stream);
  if (g_bwd_fused == 1) return plb_launch_attn_bwd_fused(p, stream);
  if (p->S <= 384) return launch_attn_bwd_split(p, stream);
  const int items = p->B * p->NH, rounds = (items + 255) / 256;
  if (items * 10 >= rounds * 256 * 9) return plb_launch_attn_bwd_fused(p, stream);
  // hybrid: whole samples worth (nearly) full rounds to the fused kernel, the rest to the two kernels
  const int per_round = 256 / p->NH;                      // samples whose (batch, head) items fit one round
  const int full = per_round > 0 ? (p->B / per_round) : 0; // full rounds available
  const int nb_f = full * per_round, rest = p->B - nb_f;
  if (hybrid_on && full >= 1 && per_round * p->NH * 10 >= 256 * 9 && rest * p->NH <= 160) {
    const PlbAttn a = attn_samples(p, 0, nb_f);
    const int rc = plb_launch_attn_bwd_fused(&a, stream);
    if (rc || rest == 0) return rc;
    const PlbAttn b = attn_samples(p, nb_f, rest);
    return launch_attn_bwd_split(&b, stream);
  }
  return launch_attn_bwd_split(p, stream);
}
