// HBM-bound row kernels of the PL-BERT step, gfx950: embedding gather + LayerNorm (fwd/bwd),
// LayerNorm(H) fwd/bwd, column sums (bias / LayerNorm-affine gradients), masked-row gather/scatter,
// masked cross-entropy fwd+bwd, slab reduction, AdamW, casts and transposes.
// One wave per row, 8-byte (4 x bf16) accesses, wave-shuffle reductions, fp32 statistics
// (layer_norm_eps = 1e-12 is below bf16 resolution).
#include "common.h"
#include "plbert_kernels.h"

namespace {

// ------------------------------------------------------------------------- embeddings (A4 / A11)
// AlbertEmbeddings.forward (modeling_albert.py:67-106): LN_E(word[id] + type[0] + pos[s]); E <= 256.
template <bool BWD>
__global__ __launch_bounds__(256) void embed_kernel(PlbEmbed p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int E = p.E;
  const int c = lane * 4;  // this lane's 4 columns
  const bool act = c < E;
  float4 g4 = make_float4(0, 0, 0, 0), b4 = g4, ty = g4;
  if (act) {
    g4 = *(const float4*)(p.gamma + c);
    b4 = *(const float4*)(p.beta + c);
    ty = *(const float4*)(p.type0 + c);
  }
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  const float invE = 1.0f / (float)E;
  for (int t = blockIdx.x * 4 + wave; t < p.T; t += gridDim.x * 4) {
    long long id = p.ids[t];
    if (id < 0 || id >= p.V) id = 0;  // host validates; never index out of the table
    const int s = t % p.S;
    float x[4] = {0, 0, 0, 0};
    if (act) {
      float4 w = *(const float4*)(p.word + (size_t)id * E + c);
      float4 ps = *(const float4*)(p.pos + (size_t)s * E + c);
      x[0] = w.x + ty.x + ps.x; x[1] = w.y + ty.y + ps.y; x[2] = w.z + ty.z + ps.z; x[3] = w.w + ty.w + ps.w;
    }
    const float mean = wave_sum(x[0] + x[1] + x[2] + x[3]) * invE;
    float d[4], vs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = act ? x[j] - mean : 0.f; vs += d[j] * d[j]; }
    const float rstd = rsqrtf(wave_sum(vs) * invE + p.eps);
    if (!BWD) {
      if (act) {
        uint2 o;
        o.x = pack_bf2(d[0] * rstd * g4.x + b4.x, d[1] * rstd * g4.y + b4.y);
        o.y = pack_bf2(d[2] * rstd * g4.z + b4.z, d[3] * rstd * g4.w + b4.w);
        *(uint2*)(p.out + (size_t)t * p.ldo + c) = o;
      }
    } else {
      float dy[4] = {0, 0, 0, 0}, xh[4], dxh[4], s1 = 0.f, s2 = 0.f;
      if (act) {
        uint2 u = *(const uint2*)(p.dout + (size_t)t * p.lddo + c);
        dy[0] = bf_lo(u.x); dy[1] = bf_hi(u.x); dy[2] = bf_lo(u.y); dy[3] = bf_hi(u.y);
      }
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = d[j] * rstd;
        dxh[j] = dy[j] * gg[j];
        s1 += dxh[j]; s2 += dxh[j] * xh[j];
        dg[j] += dy[j] * xh[j]; db[j] += dy[j];
      }
      s1 = wave_sum(s1) * invE; s2 = wave_sum(s2) * invE;
      if (act)  // gradient of the pre-LayerNorm sum, consumed by embed_scatter_kernel (no atomics)
        *(float4*)(p.dx + (size_t)t * E + c) =
            make_float4(rstd * (dxh[0] - s1 - xh[0] * s2), rstd * (dxh[1] - s1 - xh[1] * s2),
                        rstd * (dxh[2] - s1 - xh[2] * s2), rstd * (dxh[3] - s1 - xh[3] * s2));
    }
  }
  if (BWD) {
    __shared__ float red[4][2][256];
    if (act) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[wave][0][c + j] = dg[j]; red[wave][1][c + j] = db[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * E; i += 256) {
      const int which = i / E, col = i % E;
      p.partials[(size_t)blockIdx.x * 2 * E + i] =
          red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
    }
  }
}

// Scatter-add of the embedding gradient without atomics (deterministic): block v < V owns word row v
// (collects the tokens whose id is v into an LDS list, then sums their dx rows in token order), block
// V + s owns position row s (sum over the batch). Row 0 of the word table is the padding row
// (nn.Embedding(padding_idx=0), modeling_albert.py:56): no gradient.
// The token range is walked in chunks of EMB_CHUNK tokens so the list fits LDS for any T (configs/config.yml:16
// is 96 x 512 = 49,152 tokens on one GPU).
constexpr int EMB_CHUNK = 32768;
__global__ __launch_bounds__(256) void embed_scatter_kernel(PlbEmbed p, int P) {
  extern __shared__ int list[];  // 4 per-wave segments of matching token indices
  __shared__ int wcnt[4];
  __shared__ float4 part[256];
  const int E = p.E, T = p.T;
  const int row = blockIdx.x;
  const int EQ = E >> 2;                                   // float4 columns per row
  const int q = threadIdx.x % EQ, grp = threadIdx.x / EQ, ngrp = 256 / EQ;  // E = 128: 32 x 8
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add = [&](int t) {
    const float4 v = *(const float4*)(p.dx + (size_t)t * E + 4 * q);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  };
  if (row < p.V) {
    // each wave scans its own quarter of the chunk (64-token pieces, round-robin) and appends the matches to its
    // own list segment: no block barrier inside the scan. Fixed order -> deterministic.
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = T < EMB_CHUNK ? T : EMB_CHUNK;
    const int cap = (chunk + 3) / 4 + 64;
    for (int c0 = 0; c0 < T && row != 0; c0 += EMB_CHUNK) {
      const int c1 = c0 + EMB_CHUNK < T ? c0 + EMB_CHUNK : T;
      int cnt = 0;
      for (int t0 = c0 + w * 64; t0 < c1; t0 += 256) {
        const int t = t0 + lane;
        const bool hit = t < c1 && p.ids[t] == row;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (hit) list[w * cap + cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = t;
        cnt += __builtin_popcountll(m);
      }
      if (lane == 0) wcnt[w] = cnt;
      __syncthreads();
      // a few ids (separator, mask) own thousands of tokens: ngrp row groups x 4 loads in flight each
      for (int ww = 0; ww < 4; ++ww) {
        const int n = wcnt[ww];
        const int* l = list + ww * cap;
        int i = grp;
        for (; i + 7 * ngrp < n; i += 8 * ngrp) {   // the mask id owns ~12 % of the tokens: its block is the launch's duration
          float4 v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = *(const float4*)(p.dx + (size_t)l[i + k * ngrp] * E + 4 * q);
          s.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
          s.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
          s.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
          s.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
        }
        for (; i + 3 * ngrp < n; i += 4 * ngrp) {
          const int t0 = l[i], t1 = l[i + ngrp], t2 = l[i + 2 * ngrp], t3 = l[i + 3 * ngrp];
          const float4 v0 = *(const float4*)(p.dx + (size_t)t0 * E + 4 * q), v1 = *(const float4*)(p.dx + (size_t)t1 * E + 4 * q);
          const float4 v2 = *(const float4*)(p.dx + (size_t)t2 * E + 4 * q), v3 = *(const float4*)(p.dx + (size_t)t3 * E + 4 * q);
          s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
          s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; i < n; i += ngrp) add(l[i]);
      }
      __syncthreads();  // the lists are rewritten by the next chunk
    }
  } else {
    const int sidx = row - p.V;  // position row: sum over the batch
    if (sidx < p.S)
      for (int t = sidx + grp * p.S; t < T; t += ngrp * p.S) add(t);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0) {
    for (int g = 1; g < ngrp; ++g) {
      const float4 v = part[g * EQ + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (row < p.V) *(float4*)(p.dword + (size_t)row * E + 4 * q) = s;
    else if (row - p.V < P) *(float4*)(p.dpos + (size_t)(row - p.V) * E + 4 * q) = s;
  }
}

// ------------------------------------------------------------------------- LayerNorm(H) (A7/A8)
// nn.LayerNorm over the last dim, biased variance, eps inside the sqrt. NCH = ceil(H/256).
// Rows are dealt to workgroups through xcd_remap: XCD x then owns the same contiguous eighth of the token rows
// as in the GEMMs and the attention kernels that produce / consume them (worth 0.05 ms per step, measured).
// A wave handles LN_R rows per iteration with all their loads issued before the first reduction:
// the kernels are pure HBM streams and one row per wave (3 x 8 B per lane) left them latency-bound.
#ifndef LN_R_ROWS
#define LN_R_ROWS 2
#endif
constexpr int LN_R = LN_R_ROWS;
// the backward keeps x, dy and three accumulator sets per lane: one row in flight per wave (more waves per SIMD)
// measured 16.5 vs 18.7 us at 16384 x 768 (tools/ln_bench.py)
#ifndef LN_RB_ROWS
#define LN_RB_ROWS 1
#endif
constexpr int LN_RB = LN_RB_ROWS;
#ifndef LN_WIDE
#define LN_WIDE 1   // 16-byte kernels for H = 768 / 1024 (0: the 8-byte kernels for every width)
#endif
#ifndef LNW_FWD_BLOCKS
#define LNW_FWD_BLOCKS 2048
#endif

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(PlbLayerNorm p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = p.H;
  const float invH = 1.0f / (float)H;
  float4 g[NCH], b[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 4;
    g[i] = (c < H) ? *(const float4*)(p.gamma + c) : make_float4(0, 0, 0, 0);
    b[i] = (c < H) ? *(const float4*)(p.beta + c) : make_float4(0, 0, 0, 0);
  }
  for (int t0 = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * LN_R; t0 < p.T; t0 += gridDim.x * 4 * LN_R) {
    uint2 u[LN_R][NCH];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const int t = (t0 + r < p.T) ? t0 + r : p.T - 1;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        u[r][i] = (c < H) ? *(const uint2*)(p.x + (size_t)t * p.ldx + c) : make_uint2(0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const int t = t0 + r;
      float x[NCH][4], s = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        x[i][0] = bf_lo(u[r][i].x); x[i][1] = bf_hi(u[r][i].x); x[i][2] = bf_lo(u[r][i].y); x[i][3] = bf_hi(u[r][i].y);
        s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
      }
      const float mean = wave_sum(s) * invH;
      float vs = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[i][j] = (c < H) ? x[i][j] - mean : 0.f; vs += x[i][j] * x[i][j]; }
      }
      const float rstd = rsqrtf(wave_sum(vs) * invH + p.eps);
      if (t < p.T) {
        if (lane == 0 && p.mean) { p.mean[t] = mean; p.rstd[t] = rstd; }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = (lane + 64 * i) * 4;
          if (c < H) {
            uint2 o;
            o.x = pack_bf2(x[i][0] * rstd * g[i].x + b[i].x, x[i][1] * rstd * g[i].y + b[i].y);
            o.y = pack_bf2(x[i][2] * rstd * g[i].z + b[i].z, x[i][3] * rstd * g[i].w + b[i].w);
            *(uint2*)(p.y + (size_t)t * p.ldy + c) = o;
          }
        }
      }
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(PlbLayerNorm p) {
  __shared__ float red[4][3][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = p.H;
  const float invH = 1.0f / (float)H;
  float dg[NCH][4], db[NCH][4], gm[NCH][4], dxs[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 4;
    float4 g = (c < H) ? *(const float4*)(p.gamma + c) : make_float4(0, 0, 0, 0);
    gm[i][0] = g.x; gm[i][1] = g.y; gm[i][2] = g.z; gm[i][3] = g.w;
#pragma unroll
    for (int j = 0; j < 4; ++j) dg[i][j] = db[i][j] = dxs[i][j] = 0.f;
  }
  for (int t0 = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * LN_RB; t0 < p.T; t0 += gridDim.x * 4 * LN_RB) {
    uint2 ux[LN_RB][NCH], ud[LN_RB][NCH];
    float mean[LN_RB], rstd[LN_RB];
#pragma unroll
    for (int r = 0; r < LN_RB; ++r) {
      const int t = (t0 + r < p.T) ? t0 + r : p.T - 1;
      mean[r] = p.mean[t]; rstd[r] = p.rstd[t];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        ux[r][i] = (c < H) ? *(const uint2*)(p.x + (size_t)t * p.ldx + c) : make_uint2(0, 0);
        ud[r][i] = (c < H) ? *(const uint2*)(p.dy + (size_t)t * p.lddy + c) : make_uint2(0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < LN_RB; ++r) {
      const int t = t0 + r;
      if (t >= p.T) continue;  // wave-uniform
      float xh[NCH][4], dxh[NCH][4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        const float xv[4] = {bf_lo(ux[r][i].x), bf_hi(ux[r][i].x), bf_lo(ux[r][i].y), bf_hi(ux[r][i].y)};
        const float dyv[4] = {bf_lo(ud[r][i].x), bf_hi(ud[r][i].x), bf_lo(ud[r][i].y), bf_hi(ud[r][i].y)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (c < H) ? (xv[j] - mean[r]) * rstd[r] : 0.f;
          dxh[i][j] = dyv[j] * gm[i][j];
          s1 += dxh[i][j]; s2 += dxh[i][j] * xh[i][j];
          dg[i][j] += dyv[j] * xh[i][j]; db[i][j] += dyv[j];
        }
      }
      s1 = wave_sum(s1) * invH; s2 = wave_sum(s2) * invH;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < H) {
          uint2 o;
          o.x = pack_bf2(rstd[r] * (dxh[i][0] - s1 - xh[i][0] * s2), rstd[r] * (dxh[i][1] - s1 - xh[i][1] * s2));
          o.y = pack_bf2(rstd[r] * (dxh[i][2] - s1 - xh[i][2] * s2), rstd[r] * (dxh[i][3] - s1 - xh[i][3] * s2));
          *(uint2*)(p.dx + (size_t)t * p.lddx + c) = o;
          // column sums of dx as stored: the bias gradient of the Linear whose output this LayerNorm normalised
          dxs[i][0] += bf_lo(o.x); dxs[i][1] += bf_hi(o.x); dxs[i][2] += bf_lo(o.y); dxs[i][3] += bf_hi(o.y);
        }
      }
    }
  }
  // padding rows of the token dimension: keep them zero so they add nothing to the batched dW GEMMs
  for (int t = p.T + blockIdx.x * 4 + wave; t < p.Tzero; t += gridDim.x * 4) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) *(uint2*)(p.dx + (size_t)t * p.lddx + c) = make_uint2(0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[wave][0][c + j] = dg[i][j]; red[wave][1][c + j] = db[i][j]; red[wave][2][c + j] = dxs[i][j]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * H; i += 256) {  // partials[block][dgamma | dbeta | colsum(dx)]
    const int which = i / H, col = i % H;
    float* dst = &p.partials[(size_t)blockIdx.x * 3 * H + i];
    const float v = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
    *dst = p.accumulate ? *dst + v : v;
  }
}

// ---- 16-byte form for the model widths (H = 768: 96 chunks of 8 bf16 per row, H = 1024: 128) -------------------------
// A wave takes RW consecutive rows as ONE run of RW*CPR 16-byte chunks, NL = RW*CPR/64 loads per lane (lane l, load i:
// chunk l + 64 i -> row (l + 64 i) / CPR, columns ((l + 64 i) % CPR) * 8 ...): full 1-KiB wave loads and stores, half
// the memory instructions of the 8-byte form, and LNW_GROUPS row groups in flight per wave before the first
// reduction. H = 768: RW = 2, NL = 3 (load 1 straddles the two rows); H = 1024: RW = 1, NL = 2.
#ifndef LNW_GROUPS
#define LNW_GROUPS 2
#endif
template <int CPR> struct LnWide {
  static constexpr int RW = (CPR == 96) ? 2 : 1;
  static constexpr int NL = RW * CPR / 64;
  static_assert(RW * CPR % 64 == 0, "row group must fill whole wave loads");
};
DEVI void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
DEVI uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

template <int CPR>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(PlbLayerNorm p) {
  constexpr int RW = LnWide<CPR>::RW, NL = LnWide<CPR>::NL, G = LNW_GROUPS, H = CPR * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invH = 1.0f / (float)H;
  const float qs = p.out8 ? p.q_scale[0] : 1.0f;
  float amax = 0.f;
  int lrow[NL], lcol[NL];
  float gm[NL][8], bt[NL][8];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int c = lane + 64 * i;
    lrow[i] = c / CPR; lcol[i] = (c % CPR) * 8;
    const float4 g0 = *(const float4*)(p.gamma + lcol[i]), g1 = *(const float4*)(p.gamma + lcol[i] + 4);
    const float4 b0 = *(const float4*)(p.beta + lcol[i]), b1 = *(const float4*)(p.beta + lcol[i] + 4);
    gm[i][0] = g0.x; gm[i][1] = g0.y; gm[i][2] = g0.z; gm[i][3] = g0.w; gm[i][4] = g1.x; gm[i][5] = g1.y; gm[i][6] = g1.z; gm[i][7] = g1.w;
    bt[i][0] = b0.x; bt[i][1] = b0.y; bt[i][2] = b0.z; bt[i][3] = b0.w; bt[i][4] = b1.x; bt[i][5] = b1.y; bt[i][6] = b1.z; bt[i][7] = b1.w;
  }
  const int step = gridDim.x * 4 * RW * G;
  for (int t0 = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * RW * G; t0 < p.T; t0 += step) {
    uint4 u[G][NL];
#pragma unroll
    for (int gI = 0; gI < G; ++gI)
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int t = t0 + gI * RW + lrow[i];
        t = t < p.T ? t : p.T - 1;
        u[gI][i] = *(const uint4*)(p.x + (size_t)t * p.ldx + lcol[i]);
      }
#pragma unroll
    for (int gI = 0; gI < G; ++gI) {
      float x[NL][8], rs[RW], mean[RW], rstd[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) rs[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        unpack8(u[gI][i], x[i]);
        const float s = ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3])) + ((x[i][4] + x[i][5]) + (x[i][6] + x[i][7]));
#pragma unroll
        for (int r = 0; r < RW; ++r) rs[r] += (lrow[i] == r) ? s : 0.f;
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) { mean[r] = wave_sum(rs[r]) * invH; rs[r] = 0.f; }
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        float mu = mean[0];
#pragma unroll
        for (int r = 1; r < RW; ++r) mu = (lrow[i] == r) ? mean[r] : mu;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[i][j] -= mu; v += x[i][j] * x[i][j]; }
#pragma unroll
        for (int r = 0; r < RW; ++r) rs[r] += (lrow[i] == r) ? v : 0.f;
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) rstd[r] = rsqrtf(wave_sum(rs[r]) * invH + p.eps);
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int t = t0 + gI * RW + r;
        if (lane == 0 && p.mean && t < p.T) { p.mean[t] = mean[r]; p.rstd[t] = rstd[r]; }
      }
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        float rsd = rstd[0];
#pragma unroll
        for (int r = 1; r < RW; ++r) rsd = (lrow[i] == r) ? rstd[r] : rsd;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = x[i][j] * rsd * gm[i][j] + bt[i][j];
        const int t = t0 + gI * RW + lrow[i];
        if (t < p.T) {
          const uint4 pk = pack8(o);
          *(uint4*)(p.y + (size_t)t * p.ldy + lcol[i]) = pk;
          if (p.out8) {  // e4m3 image of the row as stored in bf16, for the fp8 GEMM that consumes it
            float q[8];
            unpack8(pk, q);
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(q[j]));
            uint2 w;
            w.x = pack_fp8x4(q[0] * qs, q[1] * qs, q[2] * qs, q[3] * qs, false);
            w.y = pack_fp8x4(q[4] * qs, q[5] * qs, q[6] * qs, q[7] * qs, false);
            *(uint2*)(p.out8 + (size_t)t * p.ld8 + lcol[i]) = w;
          }
        }
      }
    }
  }
  if (p.out8 && p.q_amax) {
    amax = wave_max(amax);
    if (lane == 0) atomic_max_abs(p.q_amax, amax, blockIdx.x * 4 + wave);
  }
}

template <int CPR>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(PlbLayerNorm p) {
  constexpr int RW = LnWide<CPR>::RW, NL = LnWide<CPR>::NL, H = CPR * 8;
  __shared__ float red[4][3][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invH = 1.0f / (float)H;
  const float qs = p.out8 ? p.q_scale[0] : 1.0f;
  float amax = 0.f;
  int lrow[NL], lcol[NL];
  float gm[NL][8], dg[NL][8], db[NL][8], dxs[NL][8];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int c = lane + 64 * i;
    lrow[i] = c / CPR; lcol[i] = (c % CPR) * 8;
    const float4 g0 = *(const float4*)(p.gamma + lcol[i]), g1 = *(const float4*)(p.gamma + lcol[i] + 4);
    gm[i][0] = g0.x; gm[i][1] = g0.y; gm[i][2] = g0.z; gm[i][3] = g0.w; gm[i][4] = g1.x; gm[i][5] = g1.y; gm[i][6] = g1.z; gm[i][7] = g1.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) dg[i][j] = db[i][j] = dxs[i][j] = 0.f;
  }
  const int step = gridDim.x * 4 * RW;
  int t0 = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * RW;
  // one row group ahead: the loads of group k+1 are in flight while group k is reduced and stored
  uint4 nx[NL], nd[NL];
  float nmean[NL], nrstd[NL];
#define LNW_LOAD(tb)                                                                          \
  _Pragma("unroll") for (int i = 0; i < NL; ++i) {                                            \
    int t_ = (tb) + lrow[i];                                                                  \
    t_ = t_ < p.T ? t_ : p.T - 1;                                                             \
    nx[i] = *(const uint4*)(p.x + (size_t)t_ * p.ldx + lcol[i]);                              \
    nd[i] = *(const uint4*)(p.dy + (size_t)t_ * p.lddy + lcol[i]);                            \
    nmean[i] = p.mean[t_]; nrstd[i] = p.rstd[t_];                                             \
  }
  if (t0 < p.T) { LNW_LOAD(t0); }
  for (; t0 < p.T; t0 += step) {
    uint4 ux[NL], ud[NL];
    float mean[NL], rstd[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { ux[i] = nx[i]; ud[i] = nd[i]; mean[i] = nmean[i]; rstd[i] = nrstd[i]; }
    if (t0 + step < p.T) { LNW_LOAD(t0 + step); }
    float xh[NL][8], dxh[NL][8], s1[RW], s2[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) s1[r] = s2[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float xv[8], dyv[8];
      unpack8(ux[i], xv);
      unpack8(ud[i], dyv);
      const bool live = t0 + lrow[i] < p.T;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = live ? dyv[j] : 0.f;
        xh[i][j] = (xv[j] - mean[i]) * rstd[i];
        dxh[i][j] = d * gm[i][j];
        a1 += dxh[i][j]; a2 += dxh[i][j] * xh[i][j];
        dg[i][j] += d * xh[i][j]; db[i][j] += d;
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) { s1[r] += (lrow[i] == r) ? a1 : 0.f; s2[r] += (lrow[i] == r) ? a2 : 0.f; }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) { s1[r] = wave_sum(s1[r]) * invH; s2[r] = wave_sum(s2[r]) * invH; }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float m1 = s1[0], m2 = s2[0];
#pragma unroll
      for (int r = 1; r < RW; ++r) { m1 = (lrow[i] == r) ? s1[r] : m1; m2 = (lrow[i] == r) ? s2[r] : m2; }
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rstd[i] * (dxh[i][j] - m1 - xh[i][j] * m2);
      const uint4 pk = pack8(o);
      if (t0 + lrow[i] < p.T) {
        *(uint4*)(p.dx + (size_t)(t0 + lrow[i]) * p.lddx + lcol[i]) = pk;
        float q[8];
        unpack8(pk, q);  // column sums of dx as stored
#pragma unroll
        for (int j = 0; j < 8; ++j) dxs[i][j] += q[j];
        if (p.out8) {  // e5m2 image of the gradient row for the fp8 dX GEMM
#pragma unroll
          for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(q[j]));
          uint2 w;
          w.x = pack_fp8x4(q[0] * qs, q[1] * qs, q[2] * qs, q[3] * qs, true);
          w.y = pack_fp8x4(q[4] * qs, q[5] * qs, q[6] * qs, q[7] * qs, true);
          *(uint2*)(p.out8 + (size_t)(t0 + lrow[i]) * p.ld8 + lcol[i]) = w;
        }
      }
    }
  }
#undef LNW_LOAD
  if (p.out8 && p.q_amax) {
    amax = wave_max(amax);
    if (lane == 0) atomic_max_abs(p.q_amax, amax, blockIdx.x * 4 + wave);
  }
  // padding rows of the token dimension: keep them zero so they add nothing to the batched dW GEMMs
  for (int t = p.T + blockIdx.x * 4 + wave; t < p.Tzero; t += gridDim.x * 4)
    for (int c = lane; c < CPR; c += 64) {
      *(uint4*)(p.dx + (size_t)t * p.lddx + c * 8) = make_uint4(0, 0, 0, 0);
      if (p.out8) *(uint2*)(p.out8 + (size_t)t * p.ld8 + c * 8) = make_uint2(0, 0);
    }
  // per-block partials: loads i and i' of one lane set can hold the same columns (H = 768: load 1 wraps), so the NL
  // slots are folded into the wave's LDS row one after the other
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float* r0 = &red[wave][0][lcol[i] + j];
      if (i == 0 || (CPR == 128)) {  // first touch of these columns by this wave (H = 1024: the two loads are disjoint)
        r0[0] = dg[i][j]; r0[H] = db[i][j]; r0[2 * H] = dxs[i][j];
      }
    }
    if (CPR != 128 && i + 1 < NL) __builtin_amdgcn_wave_barrier();
  }
  if (CPR != 128) {
    // H = 768: load 0 covered columns [0,512); load 1 lanes 0-31 cover [512,768) (first touch), lanes 32-63 [0,256)
    // (second touch); load 2 covers [256,768) (second touch)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float* r1 = &red[wave][0][lcol[1] + j];
      if (lane < 32) { r1[0] = dg[1][j]; r1[H] = db[1][j]; r1[2 * H] = dxs[1][j]; }
      else { r1[0] += dg[1][j]; r1[H] += db[1][j]; r1[2 * H] += dxs[1][j]; }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float* r2 = &red[wave][0][lcol[2] + j];
      r2[0] += dg[2][j]; r2[H] += db[2][j]; r2[2 * H] += dxs[2][j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * H; i += 256) {
    const int which = i / H, col = i % H;
    float* dst = &p.partials[(size_t)blockIdx.x * 3 * H + i];
    const float v = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
    *dst = p.accumulate ? *dst + v : v;
  }
}

// ------------------------------------------------------------------------------------ column sums
// scratch[split][n] = sum of rows of the split; a block covers 256 columns x its row range:
// 32 lanes x 8 columns per row, 8 rows in flight per block (4 waves x 2 half-waves).
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* X, size_t R, int N, int ld, float* scratch, int nsplit) {
  __shared__ float red[8][256];
  const int tid = threadIdx.x;
  const int rsub = tid >> 5, cl = tid & 31;
  const int c = blockIdx.x * 256 + cl * 8;
  const size_t rows_per = (R + nsplit - 1) / nsplit;
  const size_t r0 = (size_t)blockIdx.y * rows_per;
  size_t r1 = r0 + rows_per; if (r1 > R) r1 = R;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < N) {
    size_t r = r0 + rsub;
    if (BF16) {  // 4 independent 16-B loads in flight per thread (the sums are latency-bound otherwise)
      const bf16_t* px = (const bf16_t*)X + c;
      for (; r + 24 < r1; r += 32) {
        uint4 u0 = *(const uint4*)(px + r * ld), u1 = *(const uint4*)(px + (r + 8) * ld);
        uint4 u2 = *(const uint4*)(px + (r + 16) * ld), u3 = *(const uint4*)(px + (r + 24) * ld);
        acc[0] += (bf_lo(u0.x) + bf_lo(u1.x)) + (bf_lo(u2.x) + bf_lo(u3.x));
        acc[1] += (bf_hi(u0.x) + bf_hi(u1.x)) + (bf_hi(u2.x) + bf_hi(u3.x));
        acc[2] += (bf_lo(u0.y) + bf_lo(u1.y)) + (bf_lo(u2.y) + bf_lo(u3.y));
        acc[3] += (bf_hi(u0.y) + bf_hi(u1.y)) + (bf_hi(u2.y) + bf_hi(u3.y));
        acc[4] += (bf_lo(u0.z) + bf_lo(u1.z)) + (bf_lo(u2.z) + bf_lo(u3.z));
        acc[5] += (bf_hi(u0.z) + bf_hi(u1.z)) + (bf_hi(u2.z) + bf_hi(u3.z));
        acc[6] += (bf_lo(u0.w) + bf_lo(u1.w)) + (bf_lo(u2.w) + bf_lo(u3.w));
        acc[7] += (bf_hi(u0.w) + bf_hi(u1.w)) + (bf_hi(u2.w) + bf_hi(u3.w));
      }
    }
    if (!BF16) {  // fp32 partial rows (attention / GEMM epilogue column sums): 8 independent 16-B loads in flight per thread
      const float* px = (const float*)X + c;
      for (; r + 24 < r1; r += 32) {
        const float* q0 = px + r * ld;
        const float* q1 = px + (r + 8) * ld;
        const float* q2 = px + (r + 16) * ld;
        const float* q3 = px + (r + 24) * ld;
        const float4 a0 = *(const float4*)q0, b0 = *(const float4*)(q0 + 4), a1 = *(const float4*)q1, b1 = *(const float4*)(q1 + 4);
        const float4 a2 = *(const float4*)q2, b2 = *(const float4*)(q2 + 4), a3 = *(const float4*)q3, b3 = *(const float4*)(q3 + 4);
        acc[0] += (a0.x + a1.x) + (a2.x + a3.x); acc[1] += (a0.y + a1.y) + (a2.y + a3.y);
        acc[2] += (a0.z + a1.z) + (a2.z + a3.z); acc[3] += (a0.w + a1.w) + (a2.w + a3.w);
        acc[4] += (b0.x + b1.x) + (b2.x + b3.x); acc[5] += (b0.y + b1.y) + (b2.y + b3.y);
        acc[6] += (b0.z + b1.z) + (b2.z + b3.z); acc[7] += (b0.w + b1.w) + (b2.w + b3.w);
      }
    }
    for (; r < r1; r += 8) {
      if (BF16) {
        uint4 u = *(const uint4*)((const bf16_t*)X + r * ld + c);
        acc[0] += bf_lo(u.x); acc[1] += bf_hi(u.x); acc[2] += bf_lo(u.y); acc[3] += bf_hi(u.y);
        acc[4] += bf_lo(u.z); acc[5] += bf_hi(u.z); acc[6] += bf_lo(u.w); acc[7] += bf_hi(u.w);
      } else {
        const float* px = (const float*)X + r * ld + c;
        float4 a = *(const float4*)px, b = *(const float4*)(px + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rsub][cl * 8 + j] = acc[j];
  __syncthreads();
  const int col = blockIdx.x * 256 + tid;
  if (col < N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][tid];
    scratch[(size_t)blockIdx.y * N + col] = s;
  }
}

// out[i] (+)= sum over the splits, in split order (fixed: deterministic). Four slabs are loaded before the first add:
// the loop is a chain of dependent HBM reads otherwise (1.9 TB/s measured with one load in flight per thread).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* slab, int splits, size_t n, float* out, int accumulate) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 s = accumulate ? *(const float4*)(out + i) : make_float4(0, 0, 0, 0);
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      const float4 v0 = *(const float4*)(slab + (size_t)k * n + i), v1 = *(const float4*)(slab + (size_t)(k + 1) * n + i);
      const float4 v2 = *(const float4*)(slab + (size_t)(k + 2) * n + i), v3 = *(const float4*)(slab + (size_t)(k + 3) * n + i);
      s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x; s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
      s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z; s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; k < splits; ++k) {
      float4 v = *(const float4*)(slab + (size_t)k * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *(float4*)(out + i) = s;
  } else {
    for (size_t j = i; j < n; ++j) {
      float s = accumulate ? out[j] : 0.f;
      for (int k = 0; k < splits; ++k) s += slab[(size_t)k * n + j];
      out[j] = s;
    }
  }
}

// 32 columns per block (one 128-byte line per split row), 8 thread groups over the splits with 4 independent partial sums
// each: a column's sum is two dependent memory round trips deep, not nsplit / 4 (these launches sit on the side stream
// beside the weight-gradient GEMMs: at 10-15 us each, nine of them were a quarter of its busy time). Fixed order per column.
__global__ __launch_bounds__(256) void reduce_cols_kernel(const float* scratch, int nsplit, int N, int Nout, float* out,
                                                          int accumulate, int col0) {
  __shared__ float red[8][32];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < Nout) {
    const float* px = scratch + col0 + j;
    int k = g;
    for (; k + 24 < nsplit; k += 32) {
      s0 += px[(size_t)k * N]; s1 += px[(size_t)(k + 8) * N]; s2 += px[(size_t)(k + 16) * N]; s3 += px[(size_t)(k + 24) * N];
    }
    for (; k < nsplit; k += 8) s0 += px[(size_t)k * N];
  }
  red[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && j < Nout) {
    const float t = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
    out[j] = accumulate ? out[j] + t : t;
  }
}

// -------------------------------------------------------------------- masked rows + cross entropy
__global__ void gather_rows_kernel(const bf16_t* src, int lds_, const int32_t* rows, int n, int npad, int H,
                                   bf16_t* dst, int ldd) {
  const int r = blockIdx.x;
  const int chunks = H / 8;
  const bool live = r < n;
  const size_t srow = live ? (size_t)rows[r] : 0;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    uint4 v = live ? *(const uint4*)(src + srow * lds_ + c * 8) : make_uint4(0, 0, 0, 0);
    *(uint4*)(dst + (size_t)r * ldd + c * 8) = v;
  }
}
__global__ void scatter_rows_kernel(const bf16_t* src, int lds_, const int32_t* rows, int n, int H, bf16_t* dst,
                                    int ldd) {
  const int r = blockIdx.x;
  if (r >= n) return;
  const size_t drow = (size_t)rows[r];
  for (int c = threadIdx.x; c < H / 8; c += blockDim.x)
    *(uint4*)(dst + drow * ldd + c * 8) = *(const uint4*)(src + (size_t)r * lds_ + c * 8);
}
// calculate_phoneme_loss (train.py:107-131): per-sample mean over its masked indices, then mean over
// the samples that have any. One thread per sample.
__global__ void ce_prepare_kernel(const int32_t* offsets, const int32_t* flat, const int64_t* labels, int B, int S,
                                  int32_t* rows, int32_t* tgt, float* w) {
  // one block per sample; every block counts the non-empty samples itself (B is small)
  __shared__ int cnt[4];
  int c = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) c += offsets[b + 1] > offsets[b];
  c = (int)wave_sum((float)c);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  const int count = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  const int b = blockIdx.x;
  const int j0 = offsets[b], j1 = offsets[b + 1];
  const float wb = (j1 > j0) ? 1.0f / ((float)(j1 - j0) * (float)count) : 0.f;
  for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
    const int idx = flat[j];
    rows[j] = b * S + idx;
    tgt[j] = (int32_t)labels[(size_t)b * S + idx];
    w[j] = wb;
  }
}
// nn.CrossEntropyLoss on one row per wave: V <= 256 classes, 4 per lane.
__global__ __launch_bounds__(256) void ce_kernel(const float* logits, int ldl, int V, const int32_t* tgt, const float* w,
                                                 int n, int npad, float* loss_rows, bf16_t* dlogits, int ldd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= npad) return;
  const int c = lane * 4;
  if (r >= n) {  // padding rows: zero gradient
    if (c < ldd) *(uint2*)(dlogits + (size_t)r * ldd + c) = make_uint2(0, 0);
    return;
  }
  float z[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) z[j] = (c + j < V) ? logits[(size_t)r * ldl + c + j] : -INFINITY;
  const float mx = wave_max(fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3])));
  float e[4], s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { e[j] = (c + j < V) ? __expf(z[j] - mx) : 0.f; s += e[j]; }
  s = wave_sum(s);
  const int t = tgt[r];
  const float wr = w[r];
  float zt = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) zt += (c + j == t) ? z[j] : 0.f;
  zt = wave_sum(zt);
  if (lane == 0) loss_rows[r] = wr * (mx + __logf(s) - zt);
  const float inv = wr / s;
  if (c < ldd) {
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = (c + j < V) ? e[j] * inv - ((c + j == t) ? wr : 0.f) : 0.f;
    uint2 o; o.x = pack_bf2(g[0], g[1]); o.y = pack_bf2(g[2], g[3]);
    *(uint2*)(dlogits + (size_t)r * ldd + c) = o;
  }
}
// Fused token-head CE, between its two GEMM passes: one wave per row merges the per-tile (max, sum) pairs.
__global__ __launch_bounds__(256) void token_ce_combine_kernel(const float* pmax, const float* psum, int ntiles,
                                                               const float* tlogit, const int32_t* lengths, int B, int S,
                                                               int rows, float* lse, float* w, float* loss_rows) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= rows) return;
  const int T = B * S, b = t / S, sp = t - b * S;
  int len = S;
  if (t < T && lengths) { len = lengths[b]; len = len < 1 ? 1 : (len > S ? S : len); }
  if (t >= T || sp >= len) {
    if (lane == 0) { lse[t] = 0.f; w[t] = 0.f; loss_rows[t] = 0.f; }
    return;
  }
  float m = -INFINITY, l = 0.f;
  for (int i = lane; i < ntiles; i += 64) {
    const float pm = pmax[(size_t)t * ntiles + i], ps = psum[(size_t)t * ntiles + i];
    if (pm > m) { l *= __expf(m - pm); m = pm; }
    l += ps * __expf(pm - m);
  }
  const float M = wave_max(m);
  l = wave_sum(m == -INFINITY ? 0.f : l * __expf(m - M));
  if (lane == 0) {
    const float ls = M + __logf(l), wt = 1.0f / ((float)B * (float)len);
    lse[t] = ls; w[t] = wt; loss_rows[t] = wt * (ls - tlogit[t]);
  }
}
__global__ void add_scalar_kernel(float* out, const float* a, const float* b) { out[0] = a[0] + b[0]; }
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* x, int n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// -------------------------------------------------------------------------------- optimizer (A12)
// torch.optim.AdamW semantics, operation for operation (torch/optim/adamw.py -> adam.py, the default foreach form):
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// The scalars are formed on the host in DOUBLE and rounded once, as torch forms them in Python floats: 1.0f - 0.999f
// is 0.99998713e-3, not 1e-3f — a 1.3e-5 relative error in every second moment (tests/test_gpu_adamw_kernel.py).
// skip: the engine's hand-off error word (or null). Non-zero = a fused LayerNorm launch of this step timed out: its
// gradients are invalid and the whole update is left out (a uniform scalar load and branch; the host learns of it from
// plb_poll_status / plb_status without this launch having waited for anybody).
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* pb, size_t n,
                                                    float decay, float omb1, float b2, float omb2, float eps, float step,
                                                    float bc2_sqrt, float gscale, unsigned int* skip, int count_skip) {
  if (skip && *skip) {   // skip[count_skip] counts the optimizer steps left out (1: the trainable range — the host rewinds
    // its step count by it; 2: the token head, which keeps a step count of its own)
    if (count_skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skip + count_skip, 1u);
    return;
  }
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;  // n is a multiple of 4 (checked by the launcher)
  float4 P = *(const float4*)(p + i), G = *(const float4*)(g + i), M = *(const float4*)(m + i), V = *(const float4*)(v + i);
  float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gj = gg[j] * gscale;
    pp[j] *= decay;
    mm[j] = mm[j] + omb1 * (gj - mm[j]);
    vv[j] = b2 * vv[j] + omb2 * (gj * gj);
    const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
    pp[j] -= step * (mm[j] / denom);
  }
  *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
  *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
  *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
  if (pb) { uint2 o; o.x = pack_bf2(pp[0], pp[1]); o.y = pack_bf2(pp[2], pp[3]); *(uint2*)(pb + i) = o; }
}
// one thread, after the last launch of a loss call that can raise the word: this rank's count as a float, for the sum
// over the ranks (plb_launch_status_export)
__global__ void status_export_kernel(const unsigned int* ln_err, float* out) {
  const unsigned int e = *ln_err;
  *out = (float)(e < (1u << 20) ? e : (1u << 20));
}
// one thread, at the end of a loss call (plb_launch_step_status). summed: the ranks' counts after their all-reduce (or
// null): a word another rank raised becomes this rank's too — every replica skips the update, or none does
__global__ void step_status_kernel(unsigned int* ln_err, float* loss, unsigned int* host_mirror, const float* summed) {
  unsigned int e = *ln_err;
  if (summed) {
    const float f = *summed;
    if (f > 0.5f) {
      const unsigned int g = f < 1048576.f ? (unsigned int)(f + 0.5f) : (1u << 20);
      if (g > e) { e = g; *ln_err = e; }
    }
  }
  if (host_mirror) __hip_atomic_store(host_mirror, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (e && loss) *loss = __builtin_nanf("");
}
__global__ void cast_bf16_kernel(const float* src, bf16_t* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = f2bf(src[i]);
}
__global__ void transpose_cast_kernel(const float* src, int R, int C, bf16_t* dst, int ldd) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * ldd + r] = f2bf(tile[tx][k]);
  }
}
// Up to 8 transposes in ONE launch (the six weight copies the backward reads are 5 us of launch latency each after every
// optimizer step): block b belongs to the matrix whose tile range contains it.
struct TransposeMulti { const float* src[8]; bf16_t* dst[8]; int R[8], C[8], ldd[8], first[9]; int n; };
__global__ __launch_bounds__(256) void transpose_cast_multi_kernel(TransposeMulti q) {
  __shared__ float tile[32][33];
  int w = 0;
  while (w + 1 < q.n && (int)blockIdx.x >= q.first[w + 1]) ++w;
  const int R = q.R[w], C = q.C[w], ldd = q.ldd[w];
  const int tiles_c = (C + 31) / 32, b = blockIdx.x - q.first[w];
  const int c0 = (b % tiles_c) * 32, r0 = (b / tiles_c) * 32;
  const float* src = q.src[w];
  bf16_t* dst = q.dst[w];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * ldd + r] = f2bf(tile[tx][k]);
  }
}
// ------------------------------------------------------------------------------------ fp8 plumbing
// max |x| over a [rows][cols] matrix (ld elements per row) into one device scalar: per-thread 16-byte strides, wave
// shuffle, one atomic per wave. cols % 8 == 0 (bf16) / % 4 == 0 (fp32).
template <bool BF16>
__global__ __launch_bounds__(256) void amax_kernel(const void* X, size_t rows, int cols, int ld, float* out) {
  const int per = BF16 ? 8 : 4, cpr = cols / per;
  const size_t n = rows * (size_t)cpr;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cpr;
    const int c = (int)(i - r * cpr) * per;
    if (BF16) {
      const uint4 u = *(const uint4*)((const bf16_t*)X + r * ld + c);
      float f[8];
      unpack8(u, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(f[j]));
    } else {
      const float4 v = *(const float4*)((const float*)X + r * ld + c);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomic_max_abs(out, m, blockIdx.x * 4 + (threadIdx.x >> 6));
}
// Delayed scaling: the scale a tensor is quantised with in the NEXT step is fmax / (the maximum seen in this one);
// a site that saw nothing (amax 0) keeps its scale. deq = 1 / scale is what the GEMM epilogues multiply by.
// One wave per site: maximum over the site's F8_SLOTS words, then the update, then the words are cleared.
// group > 1: sites come in runs of `group` entries (the L applications of the shared layer at one operand site) that share
// ONE scale, formed from the largest maximum of the run — the weight-gradient GEMM sums products of two images over all
// applications in one launch with one dequantisation factor, so every application's image must be on the same scale.
// stats (or null): 8 floats per group — [0..3] the last four non-zero maxima of the group, [4] how many calls quantised
// values beyond the format's range by more than the top value's rounding step (this call's true maximum x the scale it was
// quantised with > 1.0625 fmt: those elements were clamped and it shows), [5] the worst such ratio, [6] maxima recorded so far (selects the history slot). Groups from hist_from on
// (the engine: every site) take their scale from the LARGEST maximum of the history instead of the last one: a step whose
// values are larger than the previous step's (a short final batch, the step after a validation pass, a loss spike, or just
// another batch) is then clamped only if it exceeds everything seen in four calls, and the clamp is counted either way.
__global__ __launch_bounds__(64) void fp8_scales_kernel(float* amax, float* scale, float* deq, int n, float fmax, int group,
                                                        int n2, float fmax2, float* stats, int hist_from, float fmt, float fmt2) {
  const int g0 = blockIdx.x * group, lane = threadIdx.x;
  if (g0 >= n) return;
  if (g0 >= n2) { fmax = fmax2; fmt = fmt2; }   // entries [n2, n): the second format's target (one launch for both halves of the site table)
  float a = 0.f;
  for (int i = g0; i < g0 + group && i < n; ++i) {
    float* w = amax + (size_t)i * F8_SLOTS * F8_STRIDE + lane * F8_STRIDE;
    a = fmaxf(a, w[0]);
    w[0] = 0.f;
  }
  a = wave_max(a);
  if (a > 0.f && a < INFINITY) {
    float target = a;
    if (stats) {
      float* st = stats + (size_t)blockIdx.x * 8;
      if (lane == 0) {
        const float used = scale[g0];               // the scale this call's values were quantised with (0 before the first)
        const float ratio = a * used / fmt;
        if (ratio > 1.0625f) { st[4] += 1.f; st[5] = fmaxf(st[5], ratio); }   // beyond the top value by more than its rounding step
        float h[4] = {st[0], st[1], st[2], st[3]};
        unsigned int* const calls = reinterpret_cast<unsigned int*>(st + 6);   // an integer count (a float would stop at 2^24 calls)
        const int slot = (int)(*calls & 3u);
        h[slot] = a;
        st[slot] = a;
        *calls += 1u;
        if ((int)blockIdx.x >= hist_from) target = fmaxf(fmaxf(h[0], h[1]), fmaxf(h[2], h[3]));
      }
      target = __shfl(target, 0);
    }
    for (int i = g0 + lane; i < g0 + group && i < n; i += 64) {
      scale[i] = fmax / target;
      deq[i] = target / fmax;
    }
  }
}
// out[r][c] = fp8(x[r][c] * scale): 8 elements per thread
template <bool BF16>
__global__ __launch_bounds__(256) void quantize_kernel(const void* X, size_t rows, int cols, int ld, const float* scale,
                                                       uint8_t* out, int ldo, int bf8) {
  const int cpr = cols / 8;
  const size_t n = rows * (size_t)cpr;
  const float s = scale[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    float f[8];
    if (BF16) {
      unpack8(*(const uint4*)((const bf16_t*)X + r * ld + c), f);
    } else {
      const float4 a = *(const float4*)((const float*)X + r * ld + c), b = *(const float4*)((const float*)X + r * ld + c + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    uint2 w;
    w.x = pack_fp8x4(f[0] * s, f[1] * s, f[2] * s, f[3] * s, bf8 != 0);
    w.y = pack_fp8x4(f[4] * s, f[5] * s, f[6] * s, f[7] * s, bf8 != 0);
    *(uint2*)(out + r * ldo + c) = w;
  }
}

// The fp8 weight copies of a training step in ONE launch (blockIdx.y = weight): quantise with the scale the previous
// quantisation's maximum gave (delayed scaling: a weight moves by <= lr per step, the clamp covers the rest) and record this
// step's maximum for the next. Contiguous matrices (ld == cols).
struct QuantMulti { const void* src[8]; uint8_t* dst[8]; const float* scale[8]; float* amax[8]; size_t n8[8]; int bf16[8]; };
__global__ __launch_bounds__(256) void quantize_multi_kernel(QuantMulti q) {
  const int w = blockIdx.y;
  const float s = q.scale[w][0];
  const size_t n = q.n8[w];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float f[8];
    if (q.bf16[w] & 1) {   // bit 0: bf16 source (else fp32); bit 1: e5m2 image (else e4m3)
      unpack8(*(const uint4*)((const bf16_t*)q.src[w] + i * 8), f);
    } else {
      const float4 a = *(const float4*)((const float*)q.src[w] + i * 8), b = *(const float4*)((const float*)q.src[w] + i * 8 + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    const bool b8 = (q.bf16[w] & 2) != 0;
    uint2 o;
    o.x = pack_fp8x4(f[0] * s, f[1] * s, f[2] * s, f[3] * s, b8);
    o.y = pack_fp8x4(f[4] * s, f[5] * s, f[6] * s, f[7] * s, b8);
    *(uint2*)(q.dst[w] + i * 8) = o;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(f[j]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomic_max_abs(q.amax[w], m, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// AlbertModel pooler (modeling_albert.py:403): one wave per output feature, fp32 dot over H, tanh.
__global__ __launch_bounds__(256) void pooler_kernel(const float* hidden, int S, int H, const float* W, const float* bias,
                                                     float* pooled) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= H) return;
  const float* x = hidden + (size_t)b * S * H;
  const float* w = W + (size_t)o * H;
  float s = 0.f;
  for (int k = lane * 4; k < H; k += 256) {
    const float4 a = *(const float4*)(x + k), c = *(const float4*)(w + k);
    s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  s = wave_sum(s);
  if (lane == 0) pooled[(size_t)b * H + o] = tanhf(s + bias[o]);
}
__global__ void bf16_to_f32_kernel(const bf16_t* src, int lds_, float* dst, int ldd, int R, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * C) return;
  const size_t r = i / C, c = i % C;
  dst[r * ldd + c] = bf2f(src[r * lds_ + c]);
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : 2)
namespace {
struct ProfScope {  // brackets every launch made while it is alive with one pair of HIP events
  int tok; hipStream_t s;
  ProfScope(int cls, hipStream_t st, double flops, double bytes) : tok(plb_prof_begin(cls, st, flops, bytes)), s(st) {}
  ~ProfScope() { plb_prof_end(tok, s); }
};
}  // namespace

extern "C" int plb_launch_embed_fwd(const PlbEmbed* p, hipStream_t stream) {
  if (p->E % 4 || p->E > 256 || p->T <= 0) return 1;
  int blocks = (p->T + 3) / 4; if (blocks > 2048) blocks = 2048;
  ProfScope ps(PLB_K_EMBED_FWD, stream, 0, (double)p->T * (8 + 2.0 * p->E));
  hipLaunchKernelGGL((embed_kernel<false>), dim3(blocks), dim3(256), 0, stream, *p);
  return LAUNCH_OK();
}
extern "C" int plb_launch_embed_scatter(const PlbEmbed* p, int P, hipStream_t stream) {
  if ((p->E != 64 && p->E != 128 && p->E != 256) || p->T <= 0) return 1;
  ProfScope ps(PLB_K_EMBED_BWD, stream, 0, (double)p->T * p->E * 8.0);
  const int chunk = p->T < EMB_CHUNK ? p->T : EMB_CHUNK;
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(p->V + P), dim3(256), (size_t)(((chunk + 3) / 4 + 64) * 4) * 4, stream, *p, P);
  return LAUNCH_OK();
}
extern "C" int plb_launch_embed_bwd(const PlbEmbed* p, hipStream_t stream) {
  if (p->E % 4 || p->E > 256 || p->T <= 0 || p->nblocks <= 0) return 1;
  ProfScope ps(PLB_K_EMBED_BWD, stream, 0, (double)p->T * (8 + 2.0 * p->E + 8.0 * p->E));
  hipLaunchKernelGGL((embed_kernel<true>), dim3(p->nblocks), dim3(256), 0, stream, *p);
  return LAUNCH_OK();
}
extern "C" int plb_launch_ln_fwd(const PlbLayerNorm* p, hipStream_t stream) {
  if (p->H % 4 || p->H > 1024 || p->T <= 0) return 1;
  int blocks = (p->T + 4 * LN_R - 1) / (4 * LN_R); if (blocks > 4096) blocks = 4096;
  ProfScope ps(PLB_K_LN_FWD, stream, 0, (double)p->T * (4.0 * p->H + 8));
  const bool fwide_ok = (p->H == 768 || p->H == 1024) && p->ldx % 8 == 0 && p->ldy % 8 == 0;
  if (p->out8 && !fwide_ok) return 1;
  if (fwide_ok && (LN_WIDE || p->out8)) {
    const int rows_per_block = 4 * (p->H == 768 ? 2 : 1) * LNW_GROUPS;
    int wb = (p->T + rows_per_block - 1) / rows_per_block; if (wb > LNW_FWD_BLOCKS) wb = LNW_FWD_BLOCKS;
    if (p->H == 768) hipLaunchKernelGGL((ln_fwd_wide_kernel<96>), dim3(wb), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL((ln_fwd_wide_kernel<128>), dim3(wb), dim3(256), 0, stream, *p);
    return LAUNCH_OK();
  }
  const int nch = (p->H + 255) / 256;
  switch (nch) {
    case 1: hipLaunchKernelGGL((ln_fwd_kernel<1>), dim3(blocks), dim3(256), 0, stream, *p); break;
    case 2: hipLaunchKernelGGL((ln_fwd_kernel<2>), dim3(blocks), dim3(256), 0, stream, *p); break;
    case 3: hipLaunchKernelGGL((ln_fwd_kernel<3>), dim3(blocks), dim3(256), 0, stream, *p); break;
    default: hipLaunchKernelGGL((ln_fwd_kernel<4>), dim3(blocks), dim3(256), 0, stream, *p); break;
  }
  return LAUNCH_OK();
}
extern "C" int plb_launch_ln_bwd(const PlbLayerNorm* p, hipStream_t stream) {
  if (p->H % 4 || p->H > 1024 || p->T <= 0 || p->nblocks <= 0) return 1;
  ProfScope ps(PLB_K_LN_BWD, stream, 0, (double)p->T * (6.0 * p->H + 8));
  const bool wide_ok = (p->H == 768 || p->H == 1024) && p->ldx % 8 == 0 && p->lddy % 8 == 0 && p->lddx % 8 == 0;
  if (p->out8 && !wide_ok) return 1;  // the fp8 copy exists in the 16-byte kernels only
  // measured (tools/ln_bench.py): at H = 1024 the 16-byte form wins (14.2 vs 17.1 us at 8192 rows), at H = 768 its
  // straddling middle load costs more than it saves (21.5 vs 18.7 us at 16384 rows): used there only for the fp8 copy
  if (wide_ok && (p->out8 || (LN_WIDE && p->H == 1024))) {
    if (p->H == 768) hipLaunchKernelGGL((ln_bwd_wide_kernel<96>), dim3(p->nblocks), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL((ln_bwd_wide_kernel<128>), dim3(p->nblocks), dim3(256), 0, stream, *p);
    return LAUNCH_OK();
  }
  const int nch = (p->H + 255) / 256;
  switch (nch) {
    case 1: hipLaunchKernelGGL((ln_bwd_kernel<1>), dim3(p->nblocks), dim3(256), 0, stream, *p); break;
    case 2: hipLaunchKernelGGL((ln_bwd_kernel<2>), dim3(p->nblocks), dim3(256), 0, stream, *p); break;
    case 3: hipLaunchKernelGGL((ln_bwd_kernel<3>), dim3(p->nblocks), dim3(256), 0, stream, *p); break;
    default: hipLaunchKernelGGL((ln_bwd_kernel<4>), dim3(p->nblocks), dim3(256), 0, stream, *p); break;
  }
  return LAUNCH_OK();
}
extern "C" int plb_launch_reduce_slabs(const float* slab, int splits, size_t n, float* out, int accumulate,
                                       hipStream_t stream) {
  if (!n) return 0;
  const size_t threads = (n + 3) / 4;
  ProfScope ps(PLB_K_REDUCE, stream, 0, 4.0 * (double)n * (splits + 1));
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, slab, splits, n,
                     out, accumulate);
  return LAUNCH_OK();
}
extern "C" int plb_launch_colsum(const void* X, int is_bf16, size_t R, int N, int ld, float* out, int Nout,
                                 int accumulate, float* scratch, int nsplit, hipStream_t stream) {
  if (N % 8 || ld % 8 || nsplit <= 0 || Nout > N) return 1;
  ProfScope ps(PLB_K_COLSUM, stream, 0, (double)R * N * (is_bf16 ? 2 : 4));
  dim3 grid((N + 255) / 256, nsplit);
  if (is_bf16) hipLaunchKernelGGL((colsum_kernel<true>), grid, dim3(256), 0, stream, X, R, N, ld, scratch, nsplit);
  else hipLaunchKernelGGL((colsum_kernel<false>), grid, dim3(256), 0, stream, X, R, N, ld, scratch, nsplit);
  if (hipGetLastError() != hipSuccess) return 2;
  hipLaunchKernelGGL(reduce_cols_kernel, dim3((unsigned)((Nout + 31) / 32)), dim3(256), 0, stream, scratch, nsplit, N,
                     Nout, out, accumulate, 0);
  return LAUNCH_OK();
}
// Second half of a column sum whose first half (plb_launch_colsum) left [nsplit][N] partial rows in scratch:
// out[0..Nout) = sums of columns [col0, col0 + Nout). Lets one pass over a matrix feed two gradient tensors.
extern "C" int plb_launch_copy_cols(const float* scratch, int nsplit, int N, int col0, int Nout, float* out,
                                    hipStream_t stream) {
  if (nsplit <= 0 || col0 < 0 || col0 + Nout > N) return 1;
  hipLaunchKernelGGL(reduce_cols_kernel, dim3((unsigned)((Nout + 31) / 32)), dim3(256), 0, stream, scratch, nsplit, N,
                     Nout, out, 0, col0);
  return LAUNCH_OK();
}
extern "C" int plb_launch_gather_rows(const bf16_t* src, int lds_, const int32_t* rows, int n, int npad, int H,
                                      bf16_t* dst, int ldd, hipStream_t stream) {
  if (H % 8 || npad <= 0) return 1;
  ProfScope ps(PLB_K_ROWS, stream, 0, 4.0 * (double)npad * H);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(npad), dim3(128), 0, stream, src, lds_, rows, n, npad, H, dst, ldd);
  return LAUNCH_OK();
}
extern "C" int plb_launch_scatter_rows(const bf16_t* src, int lds_, const int32_t* rows, int n, int H, bf16_t* dst,
                                       int ldd, hipStream_t stream) {
  if (H % 8) return 1;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scatter_rows_kernel
, dim3(n), dim3(128), 0, stream, src, lds_, rows, n, H, dst, ldd);
  return LAUNCH_OK();
}
extern "C" int plb_launch_ce_prepare(const int32_t* offsets, const int32_t* flat, const int64_t* labels, int B, int S,
                                     int32_t* rows, int32_t* tgt, float* w, hipStream_t stream) {
  hipLaunchKernelGGL(ce_prepare_kernel, dim3(B), dim3(256), 0, stream, offsets, flat, labels, B, S, rows, tgt, w);
  return LAUNCH_OK();
}
extern "C" int plb_launch_ce_fwd_bwd(const float* logits, int ldl, int V, const int32_t* tgt, const float* w, int n,
                                     int npad, float* loss_rows, bf16_t* dlogits, int ldd, hipStream_t stream) {
  if (V > 256 || ldd > 256 || ldd % 4 || npad <= 0) return 1;
  ProfScope ps(PLB_K_CE, stream, 0, (double)npad * (4.0 * V + 2.0 * ldd));
  hipLaunchKernelGGL(ce_kernel, dim3((npad + 3) / 4), dim3(256), 0, stream, logits, ldl, V, tgt, w, n, npad, loss_rows,
                     dlogits, ldd);
  return LAUNCH_OK();
}
extern "C" int plb_launch_token_ce_combine(const float* pmax, const float* psum, int ntiles, const float* tlogit,
                                           const int32_t* lengths, int B, int S, int rows, float* lse, float* w,
                                           float* loss_rows, hipStream_t stream) {
  if (ntiles < 1 || rows < B * S) return 1;
  ProfScope ps(PLB_K_TOKEN_CE, stream, 0, (double)rows * ntiles * 8.0);
  hipLaunchKernelGGL(token_ce_combine_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, pmax, psum, ntiles, tlogit, lengths,
                     B, S, rows, lse, w, loss_rows);
  return LAUNCH_OK();
}
extern "C" int plb_launch_add_scalar(float* out, const float* a, const float* b, hipStream_t stream) {
  hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, stream, out, a, b);
  return LAUNCH_OK();
}
extern "C" int plb_launch_sum_rows(const float* x, int n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, stream, x, n, out);
  return LAUNCH_OK();
}
extern "C" int plb_launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n, double lr,
                                double beta1, double beta2, double eps, double wd, int step, double grad_scale,
                                unsigned int* skip_if_nonzero, int count_skip, hipStream_t stream) {
  if (n % 4 || step < 1) return 1;
  if (!n) return 0;
  ProfScope ps(PLB_K_ADAMW, stream, 0, (double)n * 30.0);  // p,m,v read+write, g read, bf16 copy
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, p, g, m, v, p_bf16, n,
                     (float)(1.0 - lr * wd), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)(lr / bc1), (float)sqrt(bc2), (float)grad_scale, skip_if_nonzero, count_skip);
  return LAUNCH_OK();
}
extern "C" int plb_launch_step_status(unsigned int* ln_err, float* loss, unsigned int* host_mirror, const float* summed,
                                      hipStream_t stream) {
  hipLaunchKernelGGL(step_status_kernel, dim3(1), dim3(1), 0, stream, ln_err, loss, host_mirror, summed);
  return LAUNCH_OK();
}
extern "C" int plb_launch_status_export(const unsigned int* ln_err, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(status_export_kernel, dim3(1), dim3(1), 0, stream, ln_err, out);
  return LAUNCH_OK();
}
extern "C" int plb_launch_cast_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t stream) {
  if (!n) return 0;
  ProfScope ps(PLB_K_CAST, stream, 0, 6.0 * (double)n);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
  return LAUNCH_OK();
}
extern "C" int plb_launch_transpose_cast(const float* src, int R, int C, bf16_t* dst, int ldd, hipStream_t stream) {
  ProfScope ps(PLB_K_CAST, stream, 0, 6.0 * (double)R * C);
  hipLaunchKernelGGL(transpose_cast_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, src, R, C, dst, ldd);
  return LAUNCH_OK();
}
// dst[i][c][r] = bf16(src[i][r][c]) for n <= 8 matrices (R[i] x C[i], dst rows ldd[i] elements apart) in one launch
extern "C" int plb_launch_transpose_cast_multi(int n, const float* const* src, const int* R, const int* C, bf16_t* const* dst,
                                               const int* ldd, hipStream_t stream) {
  if (n < 1 || n > 8) return 1;
  TransposeMulti q = {};
  double bytes = 0;
  q.n = n;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || R[i] < 1 || C[i] < 1 || ldd[i] < R[i]) return 1;
    q.src[i] = src[i]; q.dst[i] = dst[i]; q.R[i] = R[i]; q.C[i] = C[i]; q.ldd[i] = ldd[i];
    q.first[i + 1] = q.first[i] + ((R[i] + 31) / 32) * ((C[i] + 31) / 32);
    bytes += 6.0 * (double)R[i] * C[i];
  }
  ProfScope ps(PLB_K_CAST, stream, 0, bytes);
  hipLaunchKernelGGL(transpose_cast_multi_kernel, dim3(q.first[n]), dim3(256), 0, stream, q);
  return LAUNCH_OK();
}
extern "C" int plb_launch_amax(const void* x, int is_bf16, size_t rows, int cols, int ld, float* amax, hipStream_t stream) {
  if (!rows || cols <= 0 || cols % 8 || ld % 8) return 1;
  const size_t n = rows * (size_t)(cols / (is_bf16 ? 8 : 4));
  unsigned blocks = (unsigned)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  ProfScope ps(PLB_K_FP8, stream, 0, (double)rows * cols * (is_bf16 ? 2 : 4));
  if (is_bf16) hipLaunchKernelGGL((amax_kernel<true>), dim3(blocks), dim3(256), 0, stream, x, rows, cols, ld, amax);
  else hipLaunchKernelGGL((amax_kernel<false>), dim3(blocks), dim3(256), 0, stream, x, rows, cols, ld, amax);
  return LAUNCH_OK();
}
extern "C" int plb_launch_fp8_scales2(float* amax, float* scale, float* deq, int n, float fmax, int group, int n2, float fmax2,
                                      float* stats, int hist_from, hipStream_t stream) {
  if (n <= 0) return 0;
  if (group < 1) group = 1;
  hipLaunchKernelGGL(fp8_scales_kernel, dim3((n + group - 1) / group), dim3(64), 0, stream, amax, scale, deq, n, fmax, group, n2, fmax2,
                     stats, hist_from, 448.f, 57344.f);
  return LAUNCH_OK();
}
extern "C" int plb_launch_fp8_scales(float* amax, float* scale, float* deq, int n, float fmax, int group, hipStream_t stream) {
  return plb_launch_fp8_scales2(amax, scale, deq, n, fmax, group, n, fmax, nullptr, 0, stream);
}
extern "C" int plb_launch_quantize(const void* x, int is_bf16, size_t rows, int cols, int ld, const float* scale,
                                   uint8_t* out, int ldo, int bf8, hipStream_t stream) {
  if (!rows || cols <= 0 || cols % 8 || ld % 8 || ldo % 8) return 1;
  const size_t n = rows * (size_t)(cols / 8);
  unsigned blocks = (unsigned)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  ProfScope ps(PLB_K_FP8, stream, 0, (double)rows * cols * (is_bf16 ? 3 : 5));
  if (is_bf16) hipLaunchKernelGGL((quantize_kernel<true>), dim3(blocks), dim3(256), 0, stream, x, rows, cols, ld, scale, out, ldo, bf8);
  else hipLaunchKernelGGL((quantize_kernel<false>), dim3(blocks), dim3(256), 0, stream, x, rows, cols, ld, scale, out, ldo, bf8);
  return LAUNCH_OK();
}
// n <= 8 contiguous matrices (elements[i] % 8 == 0): dst[i] = e4m3(src[i] * scale[i][0]), amax[i] <- max |src[i]|;
// is_bf16[i]: bit 0 = the source is bf16 (else fp32), bit 1 = the image is e5m2 (gradients) instead of e4m3
extern "C" int plb_launch_quantize_multi(int n, const void* const* src, const int* is_bf16, const size_t* elements,
                                         const float* const* scale, uint8_t* const* dst, float* const* amax, hipStream_t stream) {
  if (n < 1 || n > 8) return 1;
  QuantMulti q = {};
  size_t bytes = 0;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || !scale[i] || !amax[i] || elements[i] % 8) return 1;
    q.src[i] = src[i]; q.dst[i] = dst[i]; q.scale[i] = scale[i]; q.amax[i] = amax[i]; q.n8[i] = elements[i] / 8; q.bf16[i] = is_bf16[i];
    bytes += elements[i] * ((is_bf16[i] & 1) ? 3 : 5);
  }
  ProfScope ps(PLB_K_FP8, stream, 0, (double)bytes);
  hipLaunchKernelGGL(quantize_multi_kernel, dim3(n == 1 ? 1024 : 128, n), dim3(256), 0, stream, q);   // one matrix: the whole chip
  return LAUNCH_OK();
}
extern "C" int plb_launch_pooler(const float* hidden, int B, int S, int H, const float* W, const float* bias, float* pooled,
                                 hipStream_t stream) {
  if (H % 4 || B < 1 || S < 1) return 1;
  hipLaunchKernelGGL(pooler_kernel, dim3((H + 3) / 4, B), dim3(256), 0, stream, hidden, S, H, W, bias, pooled);
  return LAUNCH_OK();
}
extern "C" int plb_launch_bf16_to_f32(const bf16_t* src, int lds_, float* dst, int ldd, int R, int C, hipStream_t stream) {
  const size_t n = (size_t)R * C;
  if (!n) return 0;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, lds_, dst, ldd, R, C);
  return LAUNCH_OK();
}
