// fp8-operand instantiations of the LayerNorm-in-epilogue forms (5 / 6) and of the gelu-derivative-stash forms (7 / 8) of
// the NT pipeline GEMM (gemm_nt_pipeline.h): what lets a call in fp8 mode (BASELINE.json configs[4]) keep the fusions of the
// bf16 path — dense and FFN-output projections with their LayerNorm (modeling_albert.py:196-200, 225-238), the two dX GEMMs
// that carry a LayerNorm backward, FFN up-projection + gelu_new (activations.py:59-66) and its backward — while every operand
// is a 1-byte image. Each launch also WRITES the 1-byte image (+ running maximum) of its output for the fp8 GEMMs that read
// it next: the following projection and, at the end of the backward, the token-major weight-gradient GEMM.
// LayerNorm forms: 128-row tiles (128x384 for N % 384 == 0, else 128x256); gelu forms: 256x256 where M allows.
// A translation unit of its own: co-compiled template variants perturb each other's register allocation.
#include "gemm_nt_pipeline.h"

namespace {
template <int V, int ACT, bool ABF8, bool NOCS = false>
void launch(const PlbGemmNT* p, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL((gemm_nt_big_kernel<V, ACT, false, true, true, ABF8, NOCS>), grid, dim3(512), 0, stream, *p);
}
}  // namespace

extern "C" int plb_launch_gemm_nt_fp8_ln(const PlbGemmNT* p_in, int mode, int a_bf8, hipStream_t stream) {
  PlbGemmNT q_ = *p_in;
  q_.ln_fault = plb_ln_fault_take();
  const PlbGemmNT* p = &q_;
  if (mode != 5 && mode != 6) return 1;
  if (p->M % 1024 || p->K % 128 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 3;
  const int tile = p->N % 384 == 0 ? 384 : p->N % 256 == 0 ? 256 : 0;
  if (!tile || p->N / tile > 4) return 3;
  if (!p->deq_a || !p->deq_b) return 1;
  if (!p->ln_gamma || !p->ln_mean || !p->ln_rstd || !p->ln_xchg || !p->ln_err || !p->C) return 1;
  if (mode == 5 && (!p->ln_beta || !p->C2)) return 1;
  if (mode == 6 && (!p->aux || !p->colpart)) return 1;
  if (p->res && (((uintptr_t)p->res & 15) || p->ldr % 8)) return 1;
  if (mode == 6 && (((uintptr_t)p->aux & 15) || p->ldaux % 8)) return 1;
  if (p->C8 && (!p->q_scale || p->ldc8 % 16 || ((uintptr_t)p->C8 & 15))) return 1;
  dim3 grid((p->M / 128) * (p->N / tile));
  const double mnk = (double)p->M * p->N * p->K;
  const double bytes = ((double)p->M * p->K + (double)p->N * p->K) + (double)p->M * p->N * 4 + (p->res ? 2.0 * p->M * p->N : 0.0) +
                       (p->C8 ? (double)p->M * p->N : 0.0);
  const int tok = plb_prof_begin(mode == 5 ? PLB_K_GEMM_NT_LNFWD_FP8 : PLB_K_GEMM_NT_LNBWD_FP8, stream, 2.0 * mnk, bytes);
  if (tile == 384) {
    if (mode == 5) { if (a_bf8) launch<3, 5, true>(p, grid, stream); else launch<3, 5, false>(p, grid, stream); }
    else { if (a_bf8) launch<3, 6, true>(p, grid, stream); else launch<3, 6, false>(p, grid, stream); }
  } else {
    if (mode == 5) { if (a_bf8) launch<1, 5, true>(p, grid, stream); else launch<1, 5, false>(p, grid, stream); }
    else { if (a_bf8) launch<1, 6, true>(p, grid, stream); else launch<1, 6, false>(p, grid, stream); }
  }
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Forward: C = gelu_new'(u) (lane-layout stash, bf16), C2 = gelu_new(u) (bf16, optional), C8 = its e4m3 image. Backward:
// C = (A·B^T) * aux (bf16, optional), C8 = its e5m2 image, colpart = column-sum partials (2 rows per row tile).
// Tiles: 256x256 where M % 256 == 0 (2 rounds at 16384 x 2048 instead of 4: these launches are epilogue-bound; the tile
// fits the register file since the fp8 units are compiled with MachineSink off — build.py), else 128x256. The stash is in
// the lane layout of the tile that wrote it: forward and backward of one call pick the same tile from the same M.
extern "C" int plb_gemm_nt_fp8_gelud_tile_rows(int M) { return M % 256 == 0 ? 256 : 128; }
extern "C" int plb_launch_gemm_nt_fp8_gelud(const PlbGemmNT* p, int backward, int a_bf8, hipStream_t stream) {
  if (p->M % 128 || p->N % 256 || p->K % 128 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 3;
  if (!p->deq_a || !p->deq_b) return 1;
  if ((!backward && !p->C) || (backward && !p->aux)) return 1;
  if (!p->C8 && !(backward ? p->C : p->C2)) return 1;   // some image of the output must leave
  if (p->C8 && (!p->q_scale || p->ldc8 % 16 || ((uintptr_t)p->C8 & 15))) return 1;
  const int tm = plb_gemm_nt_fp8_gelud_tile_rows(p->M);
  dim3 grid((p->M / tm) * (p->N / 256));
  const double mnk = (double)p->M * p->N * p->K;
  const double bytes = ((double)p->M * p->K + (double)p->N * p->K) + 2.0 * p->M * p->N + ((backward ? p->C : p->C2) ? 2.0 * p->M * p->N : 0.0) +
                       (p->C8 ? (double)p->M * p->N : 0.0);
  const int tok = plb_prof_begin(backward ? PLB_K_GEMM_NT_GELUBWD_FP8 : PLB_K_GEMM_NT_GELU_FP8, stream, 2.0 * mnk, bytes);
  if (tm == 256) {
    if (backward) {
      if (a_bf8) launch<2, 8, true>(p, grid, stream); else launch<2, 8, false>(p, grid, stream);
    } else if (p->colpart) {
      if (a_bf8) launch<2, 7, true>(p, grid, stream); else launch<2, 7, false>(p, grid, stream);
    } else {
      if (a_bf8) launch<2, 7, true, true>(p, grid, stream); else launch<2, 7, false, true>(p, grid, stream);
    }
  } else if (backward) {
    if (a_bf8) launch<1, 8, true>(p, grid, stream); else launch<1, 8, false>(p, grid, stream);
  } else if (p->colpart) {
    if (a_bf8) launch<1, 7, true>(p, grid, stream); else launch<1, 7, false>(p, grid, stream);
  } else {
    if (a_bf8) launch<1, 7, true, true>(p, grid, stream); else launch<1, 7, false, true>(p, grid, stream);
  }
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
