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
#include "common.h"
#include "plbert_kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// [64 rows][64 cols] bf16, 128-B rows, chunk index XORed with (row>>1)&7: conflict-free ds_read_b128
// for 32 consecutive rows at one chunk.
DEVI int row_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
// [64 rows][64 cols] bf16 in [row/4][col/32][4][32] sub-tiles of 256 B: conflict-free tr reads.
DEVI int tr_off(int row, int col) { return (((row >> 2) << 1) + (col >> 5)) * 128 + (row & 3) * 32 + (col & 31); }

// A-operand fragment of X^T (X stored [row][col] in the tr layout): lane (m = cb*32 + (l&31), half h)
// element j <- X[rb*32 + 16s + 8(j>>2) + 4h + (j&3)][m]
DEVI bf16x8 tr_frag(const bf16_t* tile, int rb, int s, int cb, int lane) {
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3, h = g >> 1;
  const int r0 = rb * 32 + 16 * s + 4 * h + q4;
  const int c = cb * 32 + 16 * (g & 1) + 4 * p4;
  s16x4 a = lds_read_tr16(&tile[tr_off(r0, c)]);
  s16x4 b = lds_read_tr16(&tile[tr_off(r0 + 8, c)]);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// Row-layout fragment: lane (row = rb*32 + (l&31), k = ks*16 + 8h + j)
DEVI bf16x8 row_frag(const bf16_t* tile, int rb, int ks, int lane) {
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
#define ROW_FRAG(tile, rb, ks, off) (*(const bf16x8*)&(tile)[(off)[ks] + (rb) * 2048])
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
DEVI void store_transposed(const f32x16& a0, const f32x16& a1, float mult, bf16_t* patch, bf16_t* gout, int ldo,
                           int rows_valid, int lane, float* colsum = nullptr, bool accumulate = false) {
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
    if (row < rows_valid) *(uint4*)(gout + (size_t)row * ldo + c * 8) = v;
  }
  if (colsum) {
    float sacc = 0.f;
#pragma unroll 8
    for (int rr = 0; rr < 32; ++rr) sacc += (rr < rows_valid) ? bf2f(patch[rr * PS + lane]) : 0.f;
    colsum[lane] = accumulate ? colsum[lane] + sacc : sacc;
  }
}

// ---------------------------------------------------------------------------------------- forward
#ifndef FWD_WAVES
#define FWD_WAVES 3   // waves per SIMD the register allocation is held to
#endif
__global__ __launch_bounds__(256, FWD_WAVES) void attn_fwd_kernel(PlbAttn p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][64 * 64];  // [stage][K row | V tr] 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const bf16_t* qbase = p.qkv + hd * 64;
  const bf16_t* kbase = p.qkv + H + hd * 64;
  const bf16_t* vbase = p.qkv + 2 * H + hd * 64;
  const int ld = p.ldqkv;
  const int lq = lane & 31, h = lane >> 5;

  bf16x8 qf[4];
  {
    int qr = q0 + lq; qr = qr < S ? qr : S - 1;
    const bf16_t* qp = qbase + (tok0 + qr) * ld + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  const int nkt = (len + 63) >> 6;
  const int sr = tid >> 3, sc = tid & 7;  // staging: 32 rows x 8 chunks, 2 passes
  // staging registers are named scalars and every iteration loads/stores unconditionally (the
  // last one re-loads its own tile): arrays captured by lambdas were demoted to scratch.
  uint4 kr0, kr1, vr0, vr1;
  const int so_r0 = row_off(sr, sc), so_r1 = row_off(sr + 32, sc);
  const int so_t0 = tr_off(sr, sc * 8), so_t1 = tr_off(sr + 32, sc * 8);
#define KV_LOAD(kt_)                                                        \
  do {                                                                      \
    int k0_ = (kt_) * 64 + sr, k1_ = k0_ + 32;                              \
    k0_ = k0_ < S ? k0_ : S - 1; k1_ = k1_ < S ? k1_ : S - 1;               \
    kr0 = *(const uint4*)(kbase + (tok0 + k0_) * ld + sc * 8);              \
    kr1 = *(const uint4*)(kbase + (tok0 + k1_) * ld + sc * 8);              \
    vr0 = *(const uint4*)(vbase + (tok0 + k0_) * ld + sc * 8);              \
    vr1 = *(const uint4*)(vbase + (tok0 + k1_) * ld + sc * 8);              \
  } while (0)
#define KV_STORE(st_)                                                       \
  do {                                                                      \
    *(uint4*)&smem[st_][0][so_r0] = kr0; *(uint4*)&smem[st_][0][so_r1] = kr1; \
    *(uint4*)&smem[st_][1][so_t0] = vr0; *(uint4*)&smem[st_][1][so_t1] = vr1; \
  } while (0)

  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = p.scale * LOG2E;

  int kqo[4];
  row_frag_offsets(lane, kqo);
  KV_LOAD(0);
  KV_STORE(0);
  __syncthreads();
  // One tile of 64 keys out of LDS stage CUR (a literal: every LDS address below is a lane constant + an immediate; with
  // a run-time stage hipcc re-derived ~45 address VALU per tile in a VALU-bound loop). The loop runs two tiles per trip.
#define FWD_TILE(CUR, kt)                                                                                         \
  do {                                                                                                            \
    KV_LOAD((kt) + 1 < nkt ? (kt) + 1 : (kt));                                                                    \
    const bf16_t* sK = smem[CUR][0];                                                                              \
    const bf16_t* sV = smem[CUR][1];                                                                              \
    f32x16 s0 = zero16(), s1 = zero16();                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                            \
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ROW_FRAG(sK, 0, ks, kqo), qf[ks], s0, 0, 0, 0);                \
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ROW_FRAG(sK, 1, ks, kqo), qf[ks], s1, 0, 0, 0);                \
    }                                                                                                             \
    /* Softmax in the exp2 domain. The loop is VALU-bound at head_dim 64 (one v_exp per score against 256 MFMA */ \
    /* flops), so: masking only in a tile that crosses the length, the scale folded into one FMA per score (max */\
    /* taken on raw scores: the scale is positive), the row maximum as a chain of three-input maxima, and the */   \
    /* accumulator rescaled only when some query's running max actually moved (exact: alpha == 1 otherwise). */   \
    if ((kt) * 64 + 64 > len) {                                                                                   \
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
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {                                                       \
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                                  \
      l_run *= alpha;                                                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }                          \
      m_run = m_new;                                                                                              \
    }                                                                                                             \
    float ls = 0.f;                                                                                               \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                              \
      s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sl2, -m_new));                                         \
      s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sl2, -m_new));                                         \
      ls += s0[r] + s1[r];                                                                                        \
    }                                                                                                             \
    l_run += ls;                                                                                                  \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                            \
      const bf16x8 pb = acc_frag((s4 >> 1) ? s1 : s0, s4 & 1);                                                    \
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sV, s4 >> 1, s4 & 1, 0, lane), pb, o0, 0, 0, 0);       \
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sV, s4 >> 1, s4 & 1, 1, lane), pb, o1, 0, 0, 0);       \
    }                                                                                                             \
    KV_STORE((CUR) ^ 1);                                                                                          \
    __syncthreads();                                                                                              \
  } while (0)
  for (int kt = 0; kt < nkt; kt += 2) {
    FWD_TILE(0, kt);
    if (kt + 1 < nkt) FWD_TILE(1, kt + 1);
  }
#undef FWD_TILE
#undef KV_LOAD
#undef KV_STORE
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (h == 0 && q0 + lq < S)
    p.lse[((size_t)b * p.NH + hd) * S + q0 + lq] = m_run * LN2 + __logf(l_tot);
  // all waves are past the last barrier: reuse the staging LDS as per-wave transpose patches
  bf16_t* patch = &smem[0][0][0] + wave * (32 * 72);
  int rows_valid = S - q0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
  if (rows_valid > 0)
    store_transposed(o0, o1, inv, patch, p.ctx + (tok0 + q0) * p.ldctx + hd * 64, p.ldctx, rows_valid, lane);
}

// ------------------------------------------------------------------------------------- backward dQ
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(PlbAttn p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][3][64 * 64];  // [stage][K row | K tr | V row] 48 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  int qr = q0 + lq; qr = qr < S ? qr : S - 1;

  bf16x8 qf[4], dof[4];
  float delta;
  {
    const bf16_t* qp = p.qkv + hd * 64 + (tok0 + qr) * ld + 8 * h;
    const bf16_t* dop = p.dctx + (tok0 + qr) * p.lddctx + hd * 64 + 8 * h;
    const bf16_t* op = p.ctx + (tok0 + qr) * p.ldctx + hd * 64 + 8 * h;
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
  const size_t stat = ((size_t)b * p.NH + hd) * S + qr;
  if (h == 0 && q0 + lq < S) p.delta[stat] = delta;
  const float lse2 = p.lse[stat] * LOG2E;
  const float sl2 = p.scale * LOG2E;

  const int nkt = (len + 63) >> 6;
  const int sr = tid >> 3, sc = tid & 7;
  uint4 kr0, kr1, vr0, vr1;
  const int so_r0 = row_off(sr, sc), so_r1 = row_off(sr + 32, sc);
  const int so_t0 = tr_off(sr, sc * 8), so_t1 = tr_off(sr + 32, sc * 8);
#define KV_LOAD(kt_)                                                        \
  do {                                                                      \
    int k0_ = (kt_) * 64 + sr, k1_ = k0_ + 32;                              \
    k0_ = k0_ < S ? k0_ : S - 1; k1_ = k1_ < S ? k1_ : S - 1;               \
    kr0 = *(const uint4*)(kbase + (tok0 + k0_) * ld + sc * 8);              \
    kr1 = *(const uint4*)(kbase + (tok0 + k1_) * ld + sc * 8);              \
    vr0 = *(const uint4*)(vbase + (tok0 + k0_) * ld + sc * 8);              \
    vr1 = *(const uint4*)(vbase + (tok0 + k1_) * ld + sc * 8);              \
  } while (0)
#define KV_STORE(st_)                                                       \
  do {                                                                      \
    *(uint4*)&smem[st_][0][so_r0] = kr0; *(uint4*)&smem[st_][0][so_r1] = kr1; \
    *(uint4*)&smem[st_][1][so_t0] = kr0; *(uint4*)&smem[st_][1][so_t1] = kr1; \
    *(uint4*)&smem[st_][2][so_r0] = vr0; *(uint4*)&smem[st_][2][so_r1] = vr1; \
  } while (0)

  f32x16 dq0 = zero16(), dq1 = zero16();
  KV_LOAD(0);
  KV_STORE(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    KV_LOAD(kt + 1 < nkt ? kt + 1 : kt);
    const bf16_t* sK = smem[cur][0];
    const bf16_t* sKt = smem[cur][1];
    const bf16_t* sV = smem[cur][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // row constants as the initial accumulator: the query sits on the lane, so dP starts at -delta and the
      // MFMA chain leaves dP - delta ready (one VALU less per score)
      f32x16 s = zero16(), dp = splat16(-delta);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sK, kb, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sV, kb, ks, lane), dof[ks], dp, 0, 0, 0);
      }
      const int kb0 = kt * 64 + kb * 32 + 4 * h;
      if (kt * 64 + 64 > len) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb0 + (r & 3) + 8 * (r >> 2);
          s[r] = (key < len) ? s[r] : -INFINITY;   // exp2(-inf) = 0
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -lse2));
        s[r] = pr * dp[r];   // dS^T = P (dP - delta); the scale is applied once at the end
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsb = acc_frag(s, s2);
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sKt, kb, s2, 0, lane), dsb, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sKt, kb, s2, 1, lane), dsb, dq1, 0, 0, 0);
      }
    }
    KV_STORE(cur ^ 1);
    __syncthreads();
  }
#undef KV_LOAD
#undef KV_STORE
  bf16_t* patch = &smem[0][0][0] + wave * (32 * 72);
  int rows_valid = S - q0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
  // bias-gradient partial row of this wave: [(b * QT + q tile) * 4 + wave][3H], columns hd*64.. of the Q block
  float* cp = p.colpart ? p.colpart + ((size_t)(b * QT + bx) * 4 + wave) * (3 * H) + hd * 64 : nullptr;
  if (rows_valid > 0)
    store_transposed(dq0, dq1, p.scale, patch, p.dqkv + (tok0 + q0) * p.lddqkv + hd * 64, p.lddqkv, rows_valid, lane, cp,
                     p.colpart_accumulate != 0);
  else if (cp && !p.colpart_accumulate) cp[lane] = 0.f;
}

// ---------------------------------------------------------------------------------- backward dK,dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(PlbAttn p) {
  // [stage][Q row | Q tr | dO row | dO tr] + lse/delta rows
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][4][64 * 64];  // 64 KiB
  __shared__ __attribute__((aligned(16))) float sstat[2][2][64];       // [stage][lse*log2e | delta][q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int QT = (p.S + 127) >> 7;  // key tiles of one (batch, head) share Q / dO: same XCD
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = logical % QT, bh = logical / QT;
  const int hd = bh % p.NH, b = bh / p.NH;
  const int S = p.S, H = p.H;
  int len = p.lengths ? p.lengths[b] : S;
  len = len < 1 ? 1 : (len > S ? S : len);
  const int key0 = bx * 128 + wave * 32;
  const size_t tok0 = (size_t)b * S;
  const int ld = p.ldqkv;
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

  // queries past the length carry exactly zero dO in this model (no loss there), so tiles stop at len
  const int nqt = (bx * 128 < len) ? ((len + 63) >> 6) : 0;
  const int sr = tid >> 3, sc = tid & 7;
  uint4 qr0, qr1, dr0, dr1;
  float st_l = 0.f, st_d = 0.f;
  const bf16_t* qbase = p.qkv + hd * 64;
  const bf16_t* dobase = p.dctx + hd * 64;
  const size_t statb = ((size_t)b * p.NH + hd) * S;
  const int so_r0 = row_off(sr, sc), so_r1 = row_off(sr + 32, sc);
  const int so_t0 = tr_off(sr, sc * 8), so_t1 = tr_off(sr + 32, sc * 8);
  const int stq = tid & 63;  // every wave loads the 64 row statistics; wave 0 stores them
#define Q_LOAD(qt_)                                                         \
  do {                                                                      \
    int q0_ = (qt_) * 64 + sr, q1_ = q0_ + 32;                              \
    q0_ = q0_ < S ? q0_ : S - 1; q1_ = q1_ < S ? q1_ : S - 1;               \
    qr0 = *(const uint4*)(qbase + (tok0 + q0_) * ld + sc * 8);              \
    qr1 = *(const uint4*)(qbase + (tok0 + q1_) * ld + sc * 8);              \
    dr0 = *(const uint4*)(dobase + (tok0 + q0_) * p.lddctx + sc * 8);       \
    dr1 = *(const uint4*)(dobase + (tok0 + q1_) * p.lddctx + sc * 8);       \
    int qs_ = (qt_) * 64 + stq; qs_ = qs_ < S ? qs_ : S - 1;                \
    st_l = p.lse[statb + qs_] * LOG2E;                                      \
    st_d = p.delta[statb + qs_];                                            \
  } while (0)
#define Q_STORE(st_)                                                        \
  do {                                                                      \
    *(uint4*)&smem[st_][0][so_r0] = qr0; *(uint4*)&smem[st_][0][so_r1] = qr1; \
    *(uint4*)&smem[st_][1][so_t0] = qr0; *(uint4*)&smem[st_][1][so_t1] = qr1; \
    *(uint4*)&smem[st_][2][so_r0] = dr0; *(uint4*)&smem[st_][2][so_r1] = dr1; \
    *(uint4*)&smem[st_][3][so_t0] = dr0; *(uint4*)&smem[st_][3][so_t1] = dr1; \
    if (tid < 64) { sstat[st_][0][tid] = st_l; sstat[st_][1][tid] = st_d; }  \
  } while (0)

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  if (nqt > 0) { Q_LOAD(0); Q_STORE(0); }
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int cur = qt & 1;
    Q_LOAD(qt + 1 < nqt ? qt + 1 : qt);
    const bf16_t* sQ = smem[cur][0];
    const bf16_t* sQt = smem[cur][1];
    const bool need_mask = (key0 + 32 > len) || (qt * 64 + 64 > len);  // wave-uniform
    const bf16_t* sDO = smem[cur][2];
    const bf16_t* sDOt = smem[cur][3];
    // The tile body exists twice: tiles that cross the length (of keys or of queries) mask, all the others run
    // without a single compare / select — written as one select on a wave-uniform flag, hipcc if-converted the mask
    // into every tile (32 v_cmp + 32 v_cndmask + 66 scalar mask ops per tile of a VALU-bound loop).
    // Row constants as initial accumulators: dP starts at -delta[q] (q runs over the registers here), so the MFMA
    // chain leaves dP - delta.
#define DKV_TILE(MASK)                                                                                          \
  _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                            \
    f32x16 s = zero16(), dp;                                                                                    \
    float lv[16];                                                                                               \
    _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                                          \
      const int ql = qb * 32 + 8 * rg + 4 * h;                                                                  \
      const float4 l4 = *(const float4*)&sstat[cur][0][ql];                                                     \
      const float4 d4 = *(const float4*)&sstat[cur][1][ql];                                                     \
      lv[4 * rg + 0] = l4.x; lv[4 * rg + 1] = l4.y; lv[4 * rg + 2] = l4.z; lv[4 * rg + 3] = l4.w;               \
      dp[4 * rg + 0] = -d4.x; dp[4 * rg + 1] = -d4.y; dp[4 * rg + 2] = -d4.z; dp[4 * rg + 3] = -d4.w;           \
    }                                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                          \
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sQ, qb, ks, lane), kf[ks], s, 0, 0, 0);              \
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sDO, qb, ks, lane), vf[ks], dp, 0, 0, 0);           \
    }                                                                                                           \
    /* s[r] = S[q][key]: key on the lane, q = qt*64 + qb*32 + (r&3) + 8(r>>2) + 4h; dS overwrites dP */         \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
      float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -lv[r]));                                     \
      if (MASK) pr = (key_ok && (qt * 64 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < len)) ? pr : 0.f;         \
      s[r] = pr;                                                                                                \
      dp[r] = pr * dp[r];                                                                                       \
    }                                                                                                           \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                          \
      const bf16x8 pb = acc_frag(s, s2), dsb = acc_frag(dp, s2);                                                \
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sDOt, qb, s2, 0, lane), pb, dv0, 0, 0, 0);          \
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sDOt, qb, s2, 1, lane), pb, dv1, 0, 0, 0);          \
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sQt, qb, s2, 0, lane), dsb, dk0, 0, 0, 0);          \
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(sQt, qb, s2, 1, lane), dsb, dk1, 0, 0, 0);          \
    }                                                                                                           \
  }
    if (need_mask) { DKV_TILE(true) } else { DKV_TILE(false) }
#undef DKV_TILE
    Q_STORE(cur ^ 1);
    __syncthreads();
  }
#undef Q_LOAD
#undef Q_STORE
  bf16_t* patch = &smem[0][0][0] + wave * (32 * 72);
  int rows_valid = S - key0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
  float* cp = p.colpart ? p.colpart + ((size_t)(b * QT + bx) * 4 + wave) * (3 * H) + hd * 64 : nullptr;
  if (rows_valid > 0) {
    bf16_t* out = p.dqkv + (tok0 + key0) * p.lddqkv + hd * 64;
    const bool accq = p.colpart_accumulate != 0;
    store_transposed(dk0, dk1, p.scale, patch, out + H, p.lddqkv, rows_valid, lane, cp ? cp + H : nullptr, accq);
    __builtin_amdgcn_wave_barrier();
    store_transposed(dv0, dv1, 1.0f, patch, out + 2 * H, p.lddqkv, rows_valid, lane, cp ? cp + 2 * H : nullptr, accq);
  } else if (cp && !p.colpart_accumulate) {
    cp[H + lane] = 0.f;
    cp[2 * H + lane] = 0.f;
  }
}

}  // namespace

static int check_attn(const PlbAttn* p) {
  if (p->H != p->NH * 64 || p->S < 1 || p->B < 1) return 1;
  if (p->ldqkv % 8 || p->ldctx % 8) return 1;
  return 0;
}

extern "C" int plb_launch_attn_fwd(const PlbAttn* p, hipStream_t stream) {
  if (check_attn(p)) return 1;
  dim3 grid(((p->S + 127) / 128) * p->NH * p->B), block(256);
  const double unit = (double)p->B * p->NH * (double)p->S * p->S * 64.0;
  const double io = 2.0 * p->B * p->S * (double)p->H;
  const int tok = plb_prof_begin(PLB_K_ATTN_FWD, stream, 4.0 * unit, 4.0 * io);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int plb_launch_attn_bwd(const PlbAttn* p, hipStream_t stream) {
  if (check_attn(p) || p->lddctx % 8 || p->lddqkv % 8) return 1;
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
