// bf16 MFMA GEMMs of the ALBERT step, gfx950.
//
//  gemm_nt : C[M,N] = A[M,K] · B[N,K]^T (+bias) (+residual) with fused epilogues
//            forward projections (Y = X·W^T with W stored [out,in] as in the reference state-dict,
//            modeling_albert.py:166-170,196,228-230,272; model.py:28) and the dX products of the
//            backward pass (dX = dY·W computed as dY·(W^T)^T against the transposed weight copy).
//  gemm_tn : dW[N,K] = A[Mtot,N]^T · B[Mtot,K]   (reduction over tokens, split over the grid; both
//            operands are token-major, so fragments come from transposed LDS reads).
//
// Tile 128x128, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles, fp32
// accumulate. NT operands are staged by LDS-DMA (global_load_lds), TN operands through registers
// (double buffered, one barrier per k-tile); LDS images are XOR-swizzled / strip-rotated so ds_read_b128 / ds_read_b64_tr_b16 and
// the ds_write_b128 staging stores are bank-conflict free.
#include "common.h"
#include "plbert_kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

// ---- NT ------------------------------------------------------------------------------------------
// LDS image of a [128 rows][64 k] bf16 tile: 128-B rows, 16-B chunk index XORed with (row>>1)&7.
DEVI int nt_lds_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int ACT, bool OUTF32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(PlbGemmNT p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][BM * BK];  // [stage][A|B] 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int logical = xcd_remap(blockIdx.x, nwg);
  const int bm = logical / nbn, bn = logical % nbn;

  // Staging: LDS-DMA (global_load_lds, 16 B per lane). One wave-instruction fills 8 rows x 128 B of
  // the image; the destination is wave-uniform base + lane*16, so the XOR swizzle is applied to the
  // per-lane SOURCE chunk (row r, stored chunk c' <- global chunk c' ^ ((r>>1)&7)). Each wave stages
  // rows [32w, 32w+32) of A and of B: 8 DMA instructions per k-tile, no VGPR round trip.
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int drow = lane >> 3, dch = lane & 7;
  const bf16_t* gA = p.A + (size_t)(bm * BM + uw * 32 + drow) * p.lda;
  const bf16_t* gB = p.B + (size_t)(bn * BN + uw * 32 + drow) * p.ldb;
  const size_t sa = (size_t)8 * p.lda, sb = (size_t)8 * p.ldb;
  int sch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sch[i] = (dch ^ ((4 * i + (lane >> 4)) & 7)) * 8;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
#define NT_STAGE(st, kt)                                                                              \
  do {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
      __builtin_amdgcn_global_load_lds((gptr_t)(gA + i * sa + (size_t)(kt) * BK + sch[i]),            \
                                       (lptr_t)&smem[st][0][(uw * 32 + i * 8) * BK], 16, 0, 0);       \
      __builtin_amdgcn_global_load_lds((gptr_t)(gB + i * sb + (size_t)(kt) * BK + sch[i]),            \
                                       (lptr_t)&smem[st][1][(uw * 32 + i * 8) * BK], 16, 0, 0);       \
    }                                                                                                 \
  } while (0)
#define NT_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  // (row>>1)&7 of a fragment row depends on the lane only: rows are wr*64 + i*16 + frow
  const int fsw = (frow >> 1) & 7;
  const int offA = (wr * 64 + frow) * BK, offB = (wc * 64 + frow) * BK;
#define NT_COMPUTE(st)                                                                                   \
  do {                                                                                                   \
    const bf16_t* sA = smem[st][0];                                                                      \
    const bf16_t* sB = smem[st][1];                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                   \
      const int ch = ((kk * 4 + fq) ^ fsw) << 3;                                                         \
      bf16x8 af[4], bfr[4];                                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
        af[i] = *(const bf16x8*)&sA[offA + i * 16 * BK + ch];                                            \
        bfr[i] = *(const bf16x8*)&sB[offB + i * 16 * BK + ch];                                           \
      }                                                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                   \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                 \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);  \
    }                                                                                                    \
  } while (0)

  // swapped MFMA operands: D[row = n][col = m], so each lane owns 4 consecutive n of one row m
  const int nk = p.K / BK;
  NT_STAGE(0, 0);
  NT_LANDED();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) NT_STAGE(cur ^ 1, kt + 1);  // buffer cur^1 was last read before the previous barrier
    NT_COMPUTE(cur);  // one copy of the MFMA body: a peeled last iteration made the allocator shuffle AGPRs
    NT_LANDED();
    __syncthreads();
  }
#undef NT_STAGE
#undef NT_LANDED
#undef NT_COMPUTE

  // epilogue: lane holds C[m][n0..n0+3]
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = bm * BM + wr * 64 + mi * 16 + frow;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      nt_epilogue<ACT, OUTF32>(p, acc[mi][ni], m, bn * BN + wc * 64 + ni * 16 + fq * 4);
  }
}

// ---- TN ------------------------------------------------------------------------------------------
// LDS image of a [64 t][128 cols] bf16 tile for transposed reads: 16-column strips, each strip a
// run of 64 32-byte units (one per t) ordered so that t and t+8 are 4 units apart (bits 2,3 of t
// swapped) and rotated by the strip index: a half-wave's ds_read_b64_tr_b16 (8 rows x 32 B) then
// covers one contiguous 256-B window, and an 8-lane ds_write_b128 group covers 128 B.
DEVI int tn_unit(int t) { return (t & 3) | (((t >> 3) & 1) << 2) | (((t >> 2) & 1) << 3) | (t & 48); }
DEVI int tn_lds_off(int t, int col) {
  int s = col >> 4;
  return s * 1024 + (((tn_unit(t) + s) & 63) << 4) + (col & 15);
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(PlbGemmTN p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][64 * 128];  // [stage][A|B] 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int nbk = (p.K + 127) / 128;
  const int bn = blockIdx.x / nbk, bk = blockIdx.x % nbk;
  const int split = blockIdx.y;
  const int t_begin = split * p.rows_per_split;
  int t_end = t_begin + p.rows_per_split;
  if (t_end > p.Mtot) t_end = p.Mtot;
  const int nt = (t_end - t_begin) / 64;

  const int lc = tid & 15, lr = tid >> 4;  // staging: 16 rows x 16 chunks per pass, 4 passes
  const int colA = bn * 128 + lc * 8, colB = bk * 128 + lc * 8;
  const bool okA = colA < p.Ncols, okB = colB < p.K;
  uint4 ra[4], rb[4];
  auto load_tile = [&](int it) {
    const size_t t0 = (size_t)t_begin + (size_t)it * 64 + lr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = okA ? *(const uint4*)(p.A + (t0 + 16 * i) * p.lda + colA) : make_uint4(0, 0, 0, 0);
      rb[i] = okB ? *(const uint4*)(p.B + (t0 + 16 * i) * p.ldb + colB) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int off = tn_lds_off(lr + 16 * i, lc * 8);
      *(uint4*)&smem[st][0][off] = ra[i];
      *(uint4*)&smem[st][1][off] = rb[i];
    }
  };

  f32x4 acc[4][4];  // [n tile][k tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  if (nt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < nt; ++it) {
    const int cur = it & 1;
    if (it + 1 < nt) load_tile(it + 1);
    const bf16_t* sA = smem[cur][0];
    const bf16_t* sB = smem[cur][1];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int t0 = ks * 32 + 8 * g + q;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ca = wn * 64 + i * 16 + 4 * pp, cb = wk * 64 + i * 16 + 4 * pp;
        s16x4 a0 = lds_read_tr16(&sA[tn_lds_off(t0, ca)]);
        s16x4 a1 = lds_read_tr16(&sA[tn_lds_off(t0 + 4, ca)]);
        s16x4 b0 = lds_read_tr16(&sB[tn_lds_off(t0, cb)]);
        s16x4 b1 = lds_read_tr16(&sB[tn_lds_off(t0 + 4, cb)]);
        af[i] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        bfr[i] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
          // D[row = k col][col = n]: lane owns 4 consecutive k of one output row n
          acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ki], af[ni], acc[ni][ki], 0, 0, 0);
    }
    if (it + 1 < nt) store_tile(cur ^ 1);
    __syncthreads();
  }

  float* out = p.slab + (size_t)split * p.N * p.K;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = bn * 128 + wn * 64 + ni * 16 + li;
    if (n >= p.N) continue;
#pragma unroll
    for (int ki = 0; ki < 4; ++ki) {
      const int k0 = bk * 128 + wk * 64 + ki * 16 + 4 * g;
      if (k0 >= p.K) continue;
      f32x4 v = acc[ni][ki];
      *(float4*)(out + (size_t)n * p.K + k0) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

}  // namespace

static int g_nt_tile = 0;  // 0: pick per shape; 128 / 256 / 384: force that tile where the shape allows
extern "C" void plb_set_gemm_nt_tile(int tile) { g_nt_tile = tile; }
extern "C" int plb_launch_gemm_nt_big(const PlbGemmNT* p, int tile, int act, int out_f32, hipStream_t stream);

// Tile policy: fill 256 CUs. efficiency = tiles / (rounds * 256) with one workgroup per CU for the
// big tiles; 128x384 moves 4/3 the operand bytes per flop of 256x256, hence the small handicap.
static int pick_tile(const PlbGemmNT* p) {
  const bool ok256 = p->M % 256 == 0 && p->N % 256 == 0, ok384 = p->M % 128 == 0 && p->N % 384 == 0;
  const bool ok1256 = p->M % 128 == 0 && p->N % 256 == 0;
  if (g_nt_tile == 128) return 128;
  if (g_nt_tile == 256) return ok256 ? 256 : 128;
  if (g_nt_tile == 384) return ok384 ? 384 : (ok256 ? 256 : 128);
  if (g_nt_tile == 1256) return ok1256 ? 1256 : 128;
  if ((long)p->M * p->N < 256L * 256 * 64) return 128;  // small problems: more, smaller workgroups
  double e256 = 0, e384 = 0, e1256 = 0;
  if (ok256) { const long t = (long)(p->M / 256) * (p->N / 256); e256 = (double)t / (double)(((t + 255) / 256) * 256); }
  if (ok384) { const long t = (long)(p->M / 128) * (p->N / 384); e384 = 0.95 * (double)t / (double)(((t + 255) / 256) * 256); }
  if (ok1256) { const long t = (long)(p->M / 128) * (p->N / 256); e1256 = 0.85 * (double)t / (double)(((t + 255) / 256) * 256); }
  if (e256 == 0 && e384 == 0 && e1256 == 0) return 128;
  if (e1256 > e256 && e1256 > e384) return 1256;
  return e384 > e256 ? 384 : 256;
}

// Rows of column-sum partials (PlbGemmNT.colpart) the launch of this shape writes: 2 per row tile of
// the big-tile kernel picked for it, 0 when the shape runs on the 128x128 kernel (no partials there).
extern "C" int plb_gemm_nt_colpart_rows(int M, int N, int K) {
  PlbGemmNT q;
  q.M = M; q.N = N; q.K = K;
  const int tile = pick_tile(&q);
  return tile == 256 ? 2 * (M / 256) : (tile == 384 || tile == 1256) ? 2 * (M / 128) : 0;
}

extern "C" int plb_launch_gemm_nt(const PlbGemmNT* p, int act, int out_f32, hipStream_t stream) {
  if (p->M % BM || p->K % BK || p->N % 4 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 1;
  const int nbn = (p->N + BN - 1) / BN;
  dim3 grid((p->M / BM) * nbn), block(256);
  const int cls = out_f32 ? PLB_K_GEMM_NT_F32 : act == 1 ? PLB_K_GEMM_NT_GELU : act == 2 ? PLB_K_GEMM_NT_GELUBWD : PLB_K_GEMM_NT;
  const int tile = pick_tile(p);
  if (p->colpart && tile == 128) return 1;  // column-sum partials exist in the big-tile kernels only
  // algorithmic HBM bytes of the launch: each operand and each output once
  const double mnk = (double)p->M * p->N * p->K;
  const double algo_bytes = 2.0 * ((double)p->M * p->K + (double)p->N * p->K) +
                            (double)p->M * p->N * (out_f32 ? 4 : act == 1 ? 4 : 2) +
                            (p->res ? 2.0 * p->M * p->N : 0.0) + (act == 2 ? 2.0 * p->M * p->N : 0.0);
  if (tile != 128) {
    const int tokb = plb_prof_begin(cls, stream, 2.0 * mnk, algo_bytes);
    const int rc = plb_launch_gemm_nt_big(p, tile, act, out_f32, stream);
    plb_prof_end(tokb, stream);
    return rc;
  }
  // the 128x128 kernel (head, map-in and other small products; some run on the engine's side stream, where a launch
  // can sit behind the main stream's grid for longer than it runs) is a class of its own: "gemm_nt" is the pipeline kernel
  const int tok = plb_prof_begin(out_f32 ? PLB_K_GEMM_NT_F32 : PLB_K_GEMM_NT_SMALL, stream, 2.0 * mnk, algo_bytes);
  if (out_f32) {
    if (act != 0) return 1;
    hipLaunchKernelGGL((gemm_nt_kernel<0, true>), grid, block, 0, stream, *p);
  } else if (act == 0) {
    hipLaunchKernelGGL((gemm_nt_kernel<0, false>), grid, block, 0, stream, *p);
  } else if (act == 1) {
    hipLaunchKernelGGL((gemm_nt_kernel<1, false>), grid, block, 0, stream, *p);
  } else if (act == 2) {
    hipLaunchKernelGGL((gemm_nt_kernel<2, false>), grid, block, 0, stream, *p);
  } else {
    return 1;
  }
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int plb_launch_gemm_tn(const PlbGemmTN* p, hipStream_t stream) {
  if (p->Mtot % 64 || p->rows_per_split % 64 || p->K % 4 || p->N <= 0 || p->splits <= 0) return 1;
  if ((long)p->splits * p->rows_per_split < p->Mtot) return 1;
  dim3 grid(((p->N + 127) / 128) * ((p->K + 127) / 128), p->splits), block(256);
  const int tok = plb_prof_begin(PLB_K_GEMM_TN, stream, 2.0 * p->Mtot * (double)p->N * p->K,
                                 2.0 * p->Mtot * ((double)p->N + p->K) + 4.0 * p->splits * (double)p->N * p->K);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
