// Internal launch interface between the HIP kernel files and the C-ABI engine (engine.cpp).
// Not installed; the public boundary is include/plbert.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

extern "C" {

// ---- per-launch HIP-event profiler (engine.cpp). Off by default: begin() then returns -1 at once.
enum PlbKernelClass {
  PLB_K_GEMM_NT = 0, PLB_K_GEMM_NT_GELU, PLB_K_GEMM_NT_GELUBWD, PLB_K_GEMM_NT_F32, PLB_K_GEMM_TN,
  PLB_K_ATTN_FWD, PLB_K_ATTN_BWD_DQ, PLB_K_ATTN_BWD_DKV, PLB_K_LN_FWD, PLB_K_LN_BWD, PLB_K_EMBED_FWD,
  PLB_K_EMBED_BWD, PLB_K_COLSUM, PLB_K_REDUCE, PLB_K_ROWS, PLB_K_CE, PLB_K_ADAMW, PLB_K_CAST, PLB_K_TOKEN_CE, PLB_K_GEMM_NT_CE, PLB_K_GEMM_NT_SMALL, PLB_K_FP8, PLB_K_ATTN_BWD,
  PLB_K_GEMM_NT_FP8, PLB_K_GEMM_NT_GELU_FP8, PLB_K_GEMM_NT_GELUBWD_FP8,  // the fp8 launches: priced against the fp8 MFMA peak
  PLB_K_GEMM_NT_LNFWD, PLB_K_GEMM_NT_LNBWD,  // GEMM + LayerNorm epilogue (gemm_ln.hip)
  PLB_K_GEMM_NT_LNFWD_FP8, PLB_K_GEMM_NT_LNBWD_FP8, PLB_K_GEMM_TN_FP8,  // fp8 forms of the same, and of the weight-gradient GEMM
  PLB_K_NCLASS
};
int plb_prof_begin(int cls, hipStream_t s, double flops, double bytes);
void plb_prof_end(int tok, hipStream_t s);

// C[M,N] = A[M,K]·B[N,K]^T (+bias)(+res); M%128==0, K%64==0, N%4==0, B readable for ceil128(N) rows.
typedef struct {
  const bf16_t* A; int lda;
  const bf16_t* B; int ldb;
  int M, N, K;
  int Mstore;                   // rows >= Mstore are computed but not stored
  const float* bias;            // [N] or null
  const bf16_t* res; int ldr;   // residual [M,N] or null
  const bf16_t* aux; int ldaux; // act==2: pre-activation u
  bf16_t* C; int ldc;           // bf16 out (act==1: pre-activation u)
  bf16_t* C2; int ldc2;         // act==1: gelu_new(u)
  float* Cf; int ldcf;          // out_f32
  float* colpart;               // big-tile kernels only, or null: [2*M/TM][N] partial column sums of the output
  // fused GEMM + cross-entropy over a wide vocabulary (big-tile kernels, 256-column tiles; act 3 and 4):
  int ce_cols;                  // real classes: columns >= ce_cols are padding
  const int64_t* ce_tgt;        // [M] target class of each row
  float* ce_pmax; float* ce_psum;  // act 3 out: [M][N/256] per-tile row maximum / sum of exp(logit - maximum)
  float* ce_tlogit;             // act 3 out: [M] logit of the target class
  const float* ce_lse;          // act 4 in: [M] log-sum-exp of the row
  const float* ce_w;            // act 4 in: [M] row weight (0 on rows without loss)
  // fp8 operand mode (plb_launch_gemm_nt_fp8, gemm_fp8.hip): A and B are 1-byte images (lda / ldb / K count elements)
  const float* deq_a; const float* deq_b;  // device scalars: 1 / scale of each operand
  uint8_t* C8; int ldc8;        // optional fp8 copy of the output that feeds the next fp8 GEMM (act 1: of C2 = gelu; else of C)
  const float* q_scale;         // device scalar: C8 = saturate(value * q_scale[0])
  float* q_amax;                // device scalar: atomic max of |value| over the launch (next step's scale)
  int c8_bf8;                   // C8 format: 0 e4m3, 1 e5m2
  // LayerNorm fused into the epilogue (plb_launch_gemm_nt_ln, gemm_ln.hip; N = the normalised width, a row spans the
  // N / TN column tiles of its row block, which exchange their row partials in global memory inside the launch):
  const float* ln_gamma; const float* ln_beta;  // [N]
  float* ln_mean; float* ln_rstd;               // [M]: written by the forward form, read by the backward form
  float ln_eps;
  unsigned long long* ln_xchg;  // [(M/128)][N/TN producer][N/TN consumer][128 rows][2] tagged granules; zero before and after every launch
  unsigned int* ln_err;         // incremented when a hand-off timed out (results of that launch are then invalid)
  int ln_fault;                 // fault injection (plb_debug_ln_fault; tests only, 0 in every product launch): 1 = the last
                                // column tile of every row block does not publish its partials (its partners time out),
                                // 2 = column tile 0 reports a time-out and leaves the granules it consumed TAGGED (what a
                                // producer's store landing after the time-out looks like to the next launch)
} PlbGemmNT;
// LayerNorm in the epilogue of the GEMM that produces its input (big-tile kernels with 128-row tiles; M % 1024 == 0,
// N % 384 == 0 or N % 256 == 0):
//  mode 5, forward:  C = bf16(A·B^T + bias + res) ("pre", kept for the backward), C2 = LayerNorm(C)·gamma + beta,
//                    ln_mean / ln_rstd written
//  mode 6, backward: dy = bf16(A·B^T + res) is the gradient of the LayerNorm's OUTPUT and is never stored; aux = the
//                    forward's pre, ln_mean / ln_rstd read; C = the gradient of the LayerNorm's input;
//                    colpart[2*M/128][3][N] = per (row tile, wave half) dgamma | dbeta | column sums of C
// Returns 3 when the shape has no fused form (the caller runs the GEMM and the LayerNorm kernel separately).
int plb_launch_gemm_nt_ln(const PlbGemmNT* p, int mode, hipStream_t stream);
// gelu_new with the DERIVATIVE stashed: forward C = gelu_new'(A·B^T + bias), C2 = gelu_new(..); backward C = (A·B^T) * aux
// (aux = the forward's C; colpart as in plb_launch_gemm_nt). M % 256 == 0 and N % 256 == 0, else 3.
int plb_launch_gemm_nt_gelud(const PlbGemmNT* p, int backward, hipStream_t stream);
// fp8 (e4m3 weights; e4m3 or e5m2 activations / gradients) form of plb_launch_gemm_nt on the pipeline kernel.
// Returns 3 when the shape has no big-tile form (the caller then uses the bf16 GEMM).
int plb_launch_gemm_nt_fp8(const PlbGemmNT* p, int act, int a_bf8, hipStream_t stream);
// fp8-operand forms of plb_launch_gemm_nt_ln (modes 5 / 6) and plb_launch_gemm_nt_gelud (gemm_fp8_ln.hip). A / B are 1-byte
// images, deq_a / deq_b their dequantisation factors; C8 (+ q_scale, q_amax, c8_bf8) is the 1-byte image of the output the
// next fp8 GEMMs read — mode 5: of C2, mode 6: of C, gelu forward: of C2 = gelu(u), gelu backward: of C. In the gelu forms the
// bf16 image (C2 / C) may be NULL: only the fp8 image leaves. Same shape rules as the bf16 forms, K % 128 == 0; 3 = no form.
int plb_launch_gemm_nt_fp8_ln(const PlbGemmNT* p, int mode, int a_bf8, hipStream_t stream);
int plb_launch_gemm_nt_fp8_gelud(const PlbGemmNT* p, int backward, int a_bf8, hipStream_t stream);
int plb_gemm_nt_fp8_gelud_tile_rows(int M);  // rows of the tile that launcher uses at this M (colpart: 2 rows per row tile)
int plb_ln_fault_take(void);  // fault injection state shared by the LayerNorm launchers (plb_debug_ln_fault)
int plb_launch_gemm_nt(const PlbGemmNT* p, int act, int out_f32, hipStream_t stream);
int plb_gemm_nt_colpart_rows(int M, int N, int K);  // rows of colpart written for this shape (0: unsupported)
// Tuning / test hooks (not part of include/plbert.h): force a tile (0 = per-shape policy; 128, 256, 384, 1256 =
// 128x256) or a K-loop form (-1 = per-launch policy, 1 interleaved, 0 staggered) for the launches that follow.
void plb_set_gemm_nt_tile(int tile);
void plb_set_gemm_nt_prefetch(int on);

// slab[split][N][K] = A[rows,N]^T · B[rows,K] over the split's rows; Mtot%64==0, rows_per_split%64==0,
// Ncols (readable columns of A) %8==0, K%8==0.
typedef struct {
  const bf16_t* A; int lda; int Ncols;
  const bf16_t* B; int ldb;
  int Mtot, N, K;
  int rows_per_split, splits;
  float* slab;
  const float* deq_a; const float* deq_b;  // fp8 form only (plb_launch_gemm_tn_fp8): 1 / scale of the two 1-byte images
} PlbGemmTN;
int plb_launch_gemm_tn(const PlbGemmTN* p, hipStream_t stream);
// fp8 form of plb_launch_gemm_tn_big (gemm_tn_fp8.hip): A = e5m2 image [rows, Ncols], B = e4m3 image [rows, K], lda / ldb in
// bytes (multiples of 16); Ncols % 256 == 0, K % 256 == 0, rows_per_split % 128 == 0, Mtot % 128 == 0
int plb_launch_gemm_tn_fp8(const PlbGemmTN* p, hipStream_t stream);
// 256x256-tile pipeline version: Ncols % 256 == 0, K % 256 == 0 (gemm_big.hip)
int plb_launch_gemm_tn_big(const PlbGemmTN* p, hipStream_t stream);

// out[n] (+)= sum over splits of slab[s][n]   (n = count of floats)
int plb_launch_reduce_slabs(const float* slab, int splits, size_t n, float* out, int accumulate, hipStream_t stream);

// Embeddings: e = LN_E(word[ids] + type[0] + pos[t % S]) -> bf16 [T,E]
typedef struct {
  const int64_t* ids; int T, S, E, V;
  const float *word, *pos, *type0, *gamma, *beta;
  float eps;
  bf16_t* out; int ldo;
  // backward
  const bf16_t* dout; int lddo;
  float* dx;                    // fp32 [T,E]: gradient of the pre-LayerNorm embedding sum (written by embed_bwd)
  float *dword, *dpos;          // fp32 [V,E], [P,E]: written by embed_scatter (every row, no atomics)
  float* partials;              // [nblocks][2E] : dgamma | dbeta partial sums
  int nblocks;
} PlbEmbed;
int plb_launch_embed_fwd(const PlbEmbed* p, hipStream_t stream);
int plb_launch_embed_bwd(const PlbEmbed* p, hipStream_t stream);  // grid = p->nblocks; writes dx + partials
int plb_launch_embed_scatter(const PlbEmbed* p, int P, hipStream_t stream);  // dx -> dword [V,E], dpos [P,E]

// LayerNorm over rows of width H (H%4==0, H<=1024): y = LN(x)*g+b ; stats saved for the backward
typedef struct {
  const bf16_t* x; int ldx;
  const float *gamma, *beta; float eps;
  bf16_t* y; int ldy;
  float *mean, *rstd;           // [T]
  int T, H;
  int Tzero;                    // backward: rows T..Tzero-1 of dx are written as zeros
  // backward: dx = LN'(dy); partials[nblocks][3H] = dgamma | dbeta | column sums of dx (as stored)
  const bf16_t* dy; int lddy;
  bf16_t* dx; int lddx;
  float* partials; int nblocks;
  int accumulate;               // backward: add to partials instead of overwriting (the 12 applications of the shared
                                // layer sum into ONE [nblocks][3H] image: launches are ordered, so is the sum)
  // fp8 copy of the output for the fp8 GEMM that consumes it (H = 768 / 1024 kernels only): forward y8 = e4m3(y * s),
  // backward dx8 = e5m2(dx * s); q_amax collects max |value| (the next step's scale); all NULL = off
  uint8_t* out8; int ld8; const float* q_scale; float* q_amax;
} PlbLayerNorm;
// fp8 plumbing (rowops.hip). An "amax" argument anywhere in this file is one SITE: 64 words on 64-byte lines (common.h:
// F8_SLOTS x F8_STRIDE floats) that the waves of a launch spread their atomic maxima over; plb_launch_fp8_scales reduces
// n consecutive sites. |x| maximum of a bf16 / fp32 buffer into a site (atomic max; zero it first),
// scale update (delayed scaling: scale = fmax / amax, deq = 1 / scale, amax reset), quantisation of a bf16 / fp32 matrix
int plb_launch_amax(const void* x, int is_bf16, size_t rows, int cols, int ld, float* amax, hipStream_t stream);
// group: runs of `group` consecutive sites share one scale (from the largest maximum of the run); 1 = every site its own
int plb_launch_fp8_scales(float* amax, float* scale, float* deq, int n, float fmax, int group, hipStream_t stream);
// the same with a second target for entries [n2, n) (n2 a multiple of group). stats (or null): 8 floats per group — a
// four-call history of the group's maxima (groups from hist_from on take their scale from its largest entry), the number
// of calls whose values exceeded the format's range under the scale they were quantised with, the worst such ratio
int plb_launch_fp8_scales2(float* amax, float* scale, float* deq, int n, float fmax, int group, int n2, float fmax2,
                           float* stats, int hist_from, hipStream_t stream);
int plb_launch_quantize(const void* x, int is_bf16, size_t rows, int cols, int ld, const float* scale, uint8_t* out, int ldo,
                        int bf8, hipStream_t stream);
// the fp8 copies of up to 8 contiguous weight matrices in one launch: dst[i] = e4m3(src[i] * scale[i][0]); amax[i] (a site)
// collects max |src[i]| for the next step's scale (delayed scaling)
int plb_launch_quantize_multi(int n, const void* const* src, const int* is_bf16, const size_t* elements,
                              const float* const* scale, uint8_t* const* dst, float* const* amax, hipStream_t stream);
int plb_launch_ln_fwd(const PlbLayerNorm* p, hipStream_t stream);
int plb_launch_ln_bwd(const PlbLayerNorm* p, hipStream_t stream);

// out[N] (+)= column sums of X[R,N]; is_bf16 selects the element type. scratch: [nsplit][N] floats.
// Only the first Nout (<= N) sums are written to out.
int plb_launch_colsum(const void* X, int is_bf16, size_t R, int N, int ld, float* out, int Nout, int accumulate,
                      float* scratch, int nsplit, hipStream_t stream);

// out[0..Nout) = column sums [col0, col0+Nout) finished from the [nsplit][N] partial rows a plb_launch_colsum left in scratch
int plb_launch_copy_cols(const float* scratch, int nsplit, int N, int col0, int Nout, float* out, hipStream_t stream);

// pooled[b,:] = tanh(W[H,H] · hidden[b*S*H ..] + bias)  (first token of every sample, fp32)
int plb_launch_pooler(const float* hidden, int B, int S, int H, const float* W, const float* bias, float* pooled,
                      hipStream_t stream);

// Attention over the fused qkv buffer [T,3H] (Q | K | V, head h at columns h*64..h*64+63), head_dim 64.
typedef struct {
  const bf16_t* qkv; int ldqkv;
  const int32_t* lengths;       // [B] valid keys per sample, or null (= S)
  int B, S, NH, H;
  float scale;                  // head_dim^-0.5
  bf16_t* ctx; int ldctx;       // [T,H]
  float* lse;                   // [B,NH,S] MINUS the log-sum-exp of the scaled scores, in units of RAW scores (-lse/scale): the backward starts its S accumulators there
  // backward
  const bf16_t* dctx; int lddctx;
  float* delta;                 // [B,NH,S] scratch between the two backward kernels: MINUS rowsum(dO∘O)
  bf16_t* dqkv; int lddqkv;     // [T,3H]
  float* colpart;               // backward, optional: [B * ceil(S/128) * 4][3H] column sums of dqkv per (sample, 128-row tile, wave)
  int colpart_accumulate;       // add to colpart instead of overwriting (sum over the applications of the shared layer)
  // fp8 mode, optional 1-byte images of the outputs for the fp8 GEMMs that consume them (the values as rounded to bf16,
  // times *scale, saturated): forward ctx8 = e4m3(ctx) [T,H]; backward dqkv8 = e5m2(dqkv) [T,3H]. amax: the site (64 words,
  // common.h) that collects max |value| for the next step's scale. dqkv may be NULL when dqkv8 is given.
  uint8_t* ctx8; int ldctx8; const float* ctx_scale; float* ctx_amax;
  uint8_t* dqkv8; int lddqkv8; const float* dqkv_scale; float* dqkv_amax;
  // Compact-query mode (qoff != NULL; two-kernel backward only): the QUERIES of sample b are rows [qoff[b], qoff[b+1]) of
  // `q` (row stride ldq, head h at columns h*64..) — e.g. the masked positions only — while keys and values stay the S rows
  // of qkv. ctx / dctx are then indexed by the compact row ([Nq, H], Nq = nq_total = qoff[B]), lse / delta are [NH][Nq],
  // and dQ goes to dq ([Nq, lddq]; + its e5m2 image dq8 in fp8 mode) instead of dqkv's Q block; dqkv's K and V blocks
  // (and colpart, row (b * ceil(S/128) + q tile) * 4 + wave as always) are written as in the full form.
  const int32_t* qoff; const bf16_t* q; int ldq; int nq_total; bf16_t* dq; int lddq; uint8_t* dq8; int lddq8;
} PlbAttn;
int plb_launch_attn_fwd(const PlbAttn* p, hipStream_t stream);
// Default: the two-kernel form, dq (+delta) then dk,dv. plb_set_attn_bwd_fused(1) / PLBERT_ATTN_BWD=fused: ONE kernel for
// S <= 512 (attn_bwd_fused.hip: 5 products, dQ complete inside the workgroup; delta is not written)
int plb_launch_attn_bwd(const PlbAttn* p, hipStream_t stream);
int plb_launch_attn_bwd_fused(const PlbAttn* p, hipStream_t stream);
// Test hook of the gradient exchange (tests/test_gpu_comm_fake_rccl.py): the next loss call leaves out its index-th
// all-reduce piece, as a forgotten tensor would; the call must then fail in pieces_done() instead of training on a
// gradient range that was never exchanged.
void plb_debug_skip_piece(int index);
// Test hook of the LayerNorm hand-off (tests/test_gpu_handoff_fault.py): the next `launches` fused LayerNorm launches run
// with PlbGemmNT.ln_fault = mode (see there). The step they belong to must be reported (plb_poll_status / plb_status),
// must not reach the parameters (the AdamW launch skips) and must leave nothing behind once plb_status has reported it.
void plb_debug_ln_fault(int mode, int launches);
void plb_set_attn_bwd_fused(int on);

// Loss rows: row r of the gathered matrix is token rows[r]
int plb_launch_gather_rows(const bf16_t* src, int lds_, const int32_t* rows, int n, int npad, int H, bf16_t* dst,
                           int ldd, hipStream_t stream);
int plb_launch_scatter_rows(const bf16_t* src, int lds_, const int32_t* rows, int n, int H, bf16_t* dst, int ldd,
                            hipStream_t stream);
// From the CSR index lists: rows[j] = b*S + idx, tgt[j] = labels[b,idx], w[j] = 1/(n_b * count)
int plb_launch_ce_prepare(const int32_t* offsets, const int32_t* flat, const int64_t* labels, int B, int S,
                          int32_t* rows, int32_t* tgt, float* w, hipStream_t stream);
// per row: loss_rows[j] = w*(lse - z[tgt]); dlogits[j,:] = w*(softmax - onehot) (bf16, zero padded to ldd cols)
int plb_launch_ce_fwd_bwd(const float* logits, int ldl, int V, const int32_t* tgt, const float* w, int n, int npad,
                          float* loss_rows, bf16_t* dlogits, int ldd, hipStream_t stream);
// loss[0] = sum(loss_rows[0..n))  (single block, deterministic order)
int plb_launch_sum_rows(const float* x, int n, float* out, hipStream_t stream);
// Fused token-head CE, between the two GEMM passes: merge the per-tile (max, sum) of every row into lse, the row
// weight 1 / (B * len) (0 on padded positions and rows >= B*S) and the weighted loss row.
int plb_launch_token_ce_combine(const float* pmax, const float* psum, int ntiles, const float* tlogit,
                                const int32_t* lengths, int B, int S, int rows, float* lse, float* w, float* loss_rows,
                                hipStream_t stream);
int plb_launch_add_scalar(float* out, const float* a, const float* b, hipStream_t stream);

// Device-side word masking (mask.hip): labels [B,S] -> masked [B,S], counts [B], idx_padded [B,S],
// then offsets [B+1] and flat (CSR of the modified positions).
typedef struct {
  const int64_t* labels; const int32_t* lengths;  // lengths may be null (= S)
  int B, S;
  uint64_t seed; uint32_t step;
  float word_pred_prob, mask_prob, replace_prob;
  int mask_id, sep_id;
  int64_t* masked; int32_t *counts, *idx_padded, *offsets, *flat;
} PlbMask;
int plb_launch_mask(const PlbMask* p, hipStream_t stream);

// Bit-exact application of host-drawn masking decisions (mask.hip; include/plbert.h plb_apply_mask)
typedef struct {
  const int64_t* ids; const int32_t *sample_off, *word_off, *word_begin, *word_len; const int8_t* action;
  const int64_t* repl; const int64_t* word_token; int64_t sep_token; const int32_t* crop_start;
  int B, S, mask_id;
  int64_t *labels, *masked, *tokens; int32_t *lengths_out, *counts, *idx_padded, *offsets, *flat;
} PlbApplyMask;
int plb_launch_apply_mask(const PlbApplyMask* p, hipStream_t stream);

// AdamW (torch.optim.AdamW semantics) over a flat range; also refreshes the bf16 compute copy.
// skip_if_nonzero (two device words, or null): the launch leaves everything untouched when word 0 is non-zero — the engine's
// hand-off error word, so that a step whose LayerNorm statistics are invalid never reaches the parameters (no host round
// trip); with count_skip = 1 / 2 the launch then adds 1 to that word (optimizer steps left out: trainable range / token head)
int plb_launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n, double lr, double beta1,
                     double beta2, double eps, double wd, int step, double grad_scale, unsigned int* skip_if_nonzero,
                     int count_skip, hipStream_t stream);
// End of a loss call: mirror the hand-off error word into host-visible memory (host_mirror: device pointer of a pinned
// host word) and, when it is non-zero, overwrite the loss with NaN — whoever reads the loss sees that the step is invalid
int plb_launch_step_status(unsigned int* ln_err, float* loss, unsigned int* host_mirror, const float* summed, hipStream_t stream);
// *out = the word as a float (the ranks' words are summed by an all-reduce; step_status merges the sum back)
int plb_launch_status_export(const unsigned int* ln_err, float* out, hipStream_t stream);
int plb_launch_cast_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t stream);
// dst[c, r] = bf16(src[r, c]) for r<R, c<C ; dst has ldd >= R columns (zero fill is the caller's job)
int plb_launch_transpose_cast(const float* src, int R, int C, bf16_t* dst, int ldd, hipStream_t stream);
int plb_launch_transpose_cast_multi(int n, const float* const* src, const int* R, const int* C, bf16_t* const* dst,
                                    const int* ldd, hipStream_t stream);
int plb_launch_bf16_to_f32(const bf16_t* src, int lds_, float* dst, int ldd, int R, int C, hipStream_t stream);

}  // extern "C"
