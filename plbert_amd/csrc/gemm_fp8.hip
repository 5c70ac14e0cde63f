// fp8 (OCP e4m3 weights; e4m3 activations or e5m2 gradients) instantiations of the NT pipeline GEMM (BASELINE.json
// configs[4]: "fp8 (e4m3) MFMA path for QKV/FFN GEMMs"). Same kernel template as the bf16 build (gemm_nt_pipeline.h): a
// K-tile of 128-byte rows holds 128 k, one block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales replaces
// two bf16 MFMAs (twice the MFMA rate at the same LDS-DMA instruction count per K-tile row: half per FLOP), fp32
// accumulation, per-tensor dequantisation in the epilogue. A translation unit of its own so that the bf16
// instantiations keep their register allocation (co-compiled template variants perturb each other).
#include "gemm_nt_pipeline.h"

namespace {

template <int V, int ACT, bool ABF8>
int launch_fp8(const PlbGemmNT* p, hipStream_t stream) {
  constexpr int TM = (V == 2 ? 2 : 1) * 128, TN = (V == 3 ? 3 : 2) * 128;
  if (p->M % TM || p->N % TN || p->K % 128 || p->M <= 0 || p->N <= 0 || p->K <= 0) return 3;
  if (!p->deq_a || !p->deq_b || !p->C) return 1;
  if (p->C8 && (!p->q_scale || p->ldc8 % 16)) return 1;
  dim3 grid((p->M / TM) * (p->N / TN)), block(512);
  hipLaunchKernelGGL((gemm_nt_big_kernel<V, ACT, false, true, true, ABF8>), grid, block, 0, stream, *p);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Tiles: 128x384 (N % 384 == 0, plain epilogue) and 128x256. The 256x256 tile is not built in fp8: its 128
// accumulators + 96 registers of 8-register operand tuples do not fit 256 VGPRs without spilling, and at one byte per
// element the 128x256 tile already moves fewer operand bytes per flop (170 FLOP/B) than the bf16 256x256 tile (128).
template <bool ABF8>
int dispatch(const PlbGemmNT* p, int tile, int act, hipStream_t stream) {
  if (act == 0) return tile == 384 ? launch_fp8<3, 0, ABF8>(p, stream) : launch_fp8<1, 0, ABF8>(p, stream);
  if (act == 1) return launch_fp8<1, 1, ABF8>(p, stream);
  if (act == 2) return launch_fp8<1, 2, ABF8>(p, stream);
  return 1;
}

}  // namespace

extern "C" int plb_launch_gemm_nt_fp8(const PlbGemmNT* p, int act, int a_bf8, hipStream_t stream) {
  if (p->M % 128 || p->K % 128) return 3;  // odd shape: the caller runs the bf16 GEMM
  int tile;
  if (act == 0 && p->N % 384 == 0) tile = 384;
  else if (p->N % 256 == 0) tile = 1256;
  else return 3;
  const int cls = act == 1 ? PLB_K_GEMM_NT_GELU_FP8 : act == 2 ? PLB_K_GEMM_NT_GELUBWD_FP8 : PLB_K_GEMM_NT_FP8;
  const double mnk = (double)p->M * p->N * p->K;
  const double bytes = ((double)p->M * p->K + (double)p->N * p->K) + (double)p->M * p->N * (act == 1 ? 4 : 2) +
                       (p->res ? 2.0 * p->M * p->N : 0.0) + (act == 2 ? 2.0 * p->M * p->N : 0.0) +
                       (p->C8 ? (double)p->M * p->N : 0.0);
  const int tok = plb_prof_begin(cls, stream, 2.0 * mnk, bytes);
  const int rc = a_bf8 ? dispatch<true>(p, tile, act, stream) : dispatch<false>(p, tile, act, stream);
  plb_prof_end(tok, stream);
  return rc;
}
