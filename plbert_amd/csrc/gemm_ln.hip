// LayerNorm fused into the epilogue of the GEMM that produces its input (gemm_nt_pipeline.h, ACT 5 / 6): the dense and
// FFN-output projections of the shared layer (forward: AlbertAttention.LayerNorm / full_layer_layer_norm,
// modeling_albert.py:196-200, 225-238) and the two dX GEMMs whose output is the gradient of a LayerNorm's output
// (backward). A translation unit of its own so that the plain instantiations keep their register allocation.
#include "gemm_nt_pipeline.h"

static int g_ln_fault_mode = 0, g_ln_fault_launches = 0;
extern "C" void plb_debug_ln_fault(int mode, int launches) { g_ln_fault_mode = mode; g_ln_fault_launches = launches; }
extern "C" int plb_ln_fault_take(void) {
  if (g_ln_fault_launches <= 0) return 0;
  --g_ln_fault_launches;
  return g_ln_fault_mode;
}

extern "C" int plb_launch_gemm_nt_ln(const PlbGemmNT* p_in, int mode, hipStream_t stream) {
  PlbGemmNT q_ = *p_in;
  q_.ln_fault = plb_ln_fault_take();
  const PlbGemmNT* p = &q_;
  if (mode != 5 && mode != 6) return 1;
  if (p->M % 1024 || p->K % 64 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 3;   // 8 XCDs x whole row blocks
  const int tile = p->N % 384 == 0 ? 384 : p->N % 256 == 0 ? 256 : 0;
  if (!tile || p->N / tile > 4) return 3;
  if (!p->ln_gamma || !p->ln_mean || !p->ln_rstd || !p->ln_xchg || !p->ln_err || !p->C) return 1;
  if (mode == 5 && (!p->ln_beta || !p->C2)) return 1;
  if (mode == 6 && (!p->aux || !p->colpart)) return 1;
  // the epilogue's operand tiles arrive by LDS-DMA, 16 bytes per lane: rows must start on 16-byte boundaries
  if (p->res && (((uintptr_t)p->res & 15) || p->ldr % 8)) return 1;
  if (mode == 6 && (((uintptr_t)p->aux & 15) || p->ldaux % 8)) return 1;
  dim3 grid((p->M / 128) * (p->N / tile)), block(512);
  const double mnk = (double)p->M * p->N * p->K;
  const double bytes = 2.0 * ((double)p->M * p->K + (double)p->N * p->K) + (double)p->M * p->N * (mode == 5 ? 4 : 4) +
                       (p->res ? 2.0 * p->M * p->N : 0.0);
  const int tok = plb_prof_begin(mode == 5 ? PLB_K_GEMM_NT_LNFWD : PLB_K_GEMM_NT_LNBWD, stream, 2.0 * mnk, bytes);
  if (tile == 384) {
    if (mode == 5) hipLaunchKernelGGL((gemm_nt_big_kernel<3, 5, false, true>), grid, block, 0, stream, *p);
    else hipLaunchKernelGGL((gemm_nt_big_kernel<3, 6, false, true>), grid, block, 0, stream, *p);
  } else {
    if (mode == 5) hipLaunchKernelGGL((gemm_nt_big_kernel<1, 5, false, true>), grid, block, 0, stream, *p);
    else hipLaunchKernelGGL((gemm_nt_big_kernel<1, 6, false, true>), grid, block, 0, stream, *p);
  }
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// gelu_new epilogues that stash the DERIVATIVE (forms 7 / 8 of gemm_nt_pipeline.h): FFN up-projection forward
// C = gelu_new'(u), C2 = gelu_new(u) with u = A·B^T + bias (modeling_albert.py:225-232, activations.py:59-66), and the
// matching backward C = (A·B^T) * aux with aux = that stash (+ column-sum partials = the FFN bias gradient). 256x256 tiles
// (M % 256 == 0, N % 256 == 0); returns 3 for other shapes: the caller then uses forms 1 / 2 (which stash u) for BOTH.
extern "C" int plb_launch_gemm_nt_gelud(const PlbGemmNT* p, int backward, hipStream_t stream) {
  if (p->M % 256 || p->N % 256 || p->K % 64 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 3;
  if (!p->C || (!backward && !p->C2) || (backward && !p->aux)) return 1;
  dim3 grid((p->M / 256) * (p->N / 256)), block(512);
  const double mnk = (double)p->M * p->N * p->K;
  const double bytes = 2.0 * ((double)p->M * p->K + (double)p->N * p->K) + 4.0 * p->M * p->N;
  const int tok = plb_prof_begin(backward ? PLB_K_GEMM_NT_GELUBWD : PLB_K_GEMM_NT_GELU, stream, 2.0 * mnk, bytes);
  if (backward) hipLaunchKernelGGL((gemm_nt_big_kernel<2, 8, false, true>), grid, block, 0, stream, *p);
  else if (p->colpart) hipLaunchKernelGGL((gemm_nt_big_kernel<2, 7, false, true>), grid, block, 0, stream, *p);
  else hipLaunchKernelGGL((gemm_nt_big_kernel<2, 7, false, true, false, false, true>), grid, block, 0, stream, *p);
  plb_prof_end(tok, stream);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
