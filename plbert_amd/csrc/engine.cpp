// C-ABI engine of the PL-BERT hot path (include/plbert.h): owns the workspace layout and the
// launch sequence of one forward / loss+backward / AdamW step.  No device allocation, no host
// synchronisation: everything is enqueued on the caller's HIP stream, so a caller may capture a
// step into a hipGraph.
//
// Memory plan (sized for 288 GB HBM3E): every activation of all L applications of the shared layer
// is stashed in bf16 as [L][Tp][width] (Tp = tokens rounded up to 128), and so is every gradient
// that feeds a weight gradient.  Weight sharing then turns the 12 per-layer dW products of the
// reference's autograd into ONE token-major GEMM per weight with reduction length L*Tp, which is
// split over the grid into fp32 slabs and reduced in fixed order (deterministic, no atomics).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <map>
#include <new>
#include <string>

#include "../../include/plbert.h"
#include "plbert_kernels.h"

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}
extern "C" const char* plb_last_error(void) { return g_err; }

// Events that only order one HIP stream of this engine behind another: no timing, and a DEVICE-scope release when
// recorded (the default is a system-scope release, i.e. an L2 write-back for the host's benefit: nobody on the host
// reads what these events publish).
static const unsigned kStreamOrderEvent = hipEventDisableTiming | hipEventReleaseToDevice;


// ---- per-launch HIP-event profiler ----------------------------------------------------------------------
// When enabled every launcher brackets its kernel with two events on the launch stream; reading
// synchronises on them and sums elapsed time, launches and algorithmic flops/bytes per kernel class.
#include <vector>
namespace {
struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
const char* const kClassNames[PLB_K_NCLASS] = {
    "gemm_nt", "gemm_nt_gelu", "gemm_nt_gelubwd", "gemm_nt_f32", "gemm_tn", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv",
    "ln_fwd", "ln_bwd", "embed_fwd", "embed_bwd", "colsum", "reduce_slabs", "gather_scatter_rows", "cross_entropy",
    "adamw", "cast_transpose", "token_ce", "gemm_nt_ce", "gemm_nt_small", "fp8_quantize", "attn_bwd", "gemm_nt_fp8",
    "gemm_nt_gelu_fp8", "gemm_nt_gelubwd_fp8", "gemm_nt_lnfwd", "gemm_nt_lnbwd", "gemm_nt_lnfwd_fp8", "gemm_nt_lnbwd_fp8",
    "gemm_tn_fp8"};
hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
extern "C" int plb_prof_begin(int cls, hipStream_t s, double flops, double bytes) {
  if (!g_prof_on) return -1;
  ProfRec r{cls, prof_event(), prof_event(), flops, bytes};
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
  return (int)g_prof.size() - 1;
}
extern "C" void plb_prof_end(int tok, hipStream_t s) {
  if (tok >= 0 && tok < (int)g_prof.size()) (void)hipEventRecord(g_prof[tok].b, s);
}
extern "C" void plb_profile_enable(int on) { g_prof_on = on != 0; }
extern "C" int plb_profile_num_classes(void) { return PLB_K_NCLASS; }
extern "C" const char* plb_profile_class_name(int cls) { return (cls >= 0 && cls < PLB_K_NCLASS) ? kClassNames[cls] : ""; }
// Waits for every recorded launch, fills per-class totals (arrays of plb_profile_num_classes()
// entries: milliseconds, launches, flops, bytes) and clears the record.
extern "C" int plb_profile_read(double* ms, int64_t* launches, double* flops, double* bytes) {
  for (int i = 0; i < PLB_K_NCLASS; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = 0; bytes[i] = 0; }
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) return fail("plb_profile_read: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return fail("plb_profile_read: elapsed failed");
    ms[r.cls] += t; launches[r.cls] += 1; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes;
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_prof.clear();
  return 0;
}

// ---- RCCL, resolved at run time -----------------------------------------------------------------------------
// The library is not linked against RCCL: a Python host has torch's own librccl.so.1 mapped already (one RCCL per
// process), a C / C++ host gets the system one. Only the handful of entry points of the gradient exchange are bound;
// types restated from rccl.h (the NCCL API): opaque communicator, 128-byte unique id, int result / enum codes.
namespace {
struct RcclId { char internal[128]; };
typedef void* RcclComm;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
RcclApi g_rccl;
const char* rccl_load() {  // nullptr on success, else what failed
  if (g_rccl.ok) return nullptr;
  const char* env = getenv("PLBERT_RCCL_LIB");
  void* h = nullptr;
  if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // already in the process (torch's copy)
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return "cannot load librccl.so.1 (set PLBERT_RCCL_LIB)";
  g_rccl.handle = h;
#define RSYM(field, name) \
  *(void**)(&g_rccl.field) = dlsym(h, name); \
  if (!g_rccl.field) return "librccl lacks " name
  RSYM(GetUniqueId, "ncclGetUniqueId");
  RSYM(CommInitRank, "ncclCommInitRank");
  RSYM(CommDestroy, "ncclCommDestroy");
  RSYM(AllReduce, "ncclAllReduce");
  RSYM(Broadcast, "ncclBroadcast");
  RSYM(GetVersion, "ncclGetVersion");
  RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
  g_rccl.ok = true;
  return nullptr;
}
}  // namespace

namespace {

inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct Carve {  // bump allocator over the workspace, 256-B aligned regions
  int64_t off = 0;
  int64_t take(int64_t bytes) {
    int64_t o = off;
    off = rup(off + bytes, 256);
    return o;
  }
};

// ---- happens-before audit of the backward's three streams (debug: PLBERT_HB_AUDIT=1 / plb_debug_hb_audit) --------------
// A host-side MODEL of the ordering the engine asks HIP for, kept beside the real calls: every stream carries a vector
// clock; an event record snapshots the recording stream's clock, a stream wait merges the snapshot into the waiter's.
// Every access to a buffer that more than one stream touches in a loss call (a flat gradient range, the partial-row
// tables, the scratch / slab areas, ...) is logged as (byte range, stream, that stream's tick, read or write), and is
// checked on entry against every logged access of ANOTHER stream to overlapping bytes where at least one of the two
// writes: the earlier one must be inside the later stream's clock, i.e. ordered before it by a record / wait chain.
// It reasons about the calls the engine makes, not about timing: a missing hipStreamWaitEvent is reported on every
// run, not once in eighty. DESIGN.md section 4 carries the table this checks.
struct HbAudit {
  enum { MAIN = 0, SIDE = 1, COMM = 2, NS = 3 };
  typedef std::array<uint64_t, NS> VC;
  struct Acc { const char* what; uintptr_t a, b; int st; uint64_t tick; bool wr; };
  bool on = false;
  int break_wait = -1;     // test hook: the MODEL forgets its n-th wait of the next loss call (the HIP call is still made)
  int waits = 0;
  VC vc[NS] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  std::map<hipEvent_t, VC> ev;
  std::vector<Acc> log;
  int64_t checks = 0;
  int violations = 0;
  std::string first;
  static const char* name(int st) { return st == MAIN ? "main" : st == SIDE ? "side" : "comm"; }
  void record(hipEvent_t e, int st) { vc[st][st] += 1; ev[e] = vc[st]; }
  void wait(int st, hipEvent_t e) {
    const int n = waits++;
    if (n == break_wait) { break_wait = -1; return; }
    auto it = ev.find(e);
    if (it == ev.end()) return;   // never recorded: HIP treats the wait as a no-op, so does the model
    for (int i = 0; i < NS; ++i) if (it->second[i] > vc[st][i]) vc[st][i] = it->second[i];
  }
  void access(int st, const void* p, size_t bytes, bool wr, const char* what) {
    if (!p || !bytes) return;
    const uintptr_t a = (uintptr_t)p, b = a + bytes;
    vc[st][st] += 1;
    for (const Acc& x : log) {
      if (x.st == st || !(wr || x.wr) || x.b <= a || b <= x.a) continue;
      ++checks;
      if (x.tick > vc[st][x.st]) {
        if (!violations++) {
          char m[384];
          snprintf(m, sizeof(m), "%s of '%s' on the %s stream is not ordered after the %s of '%s' on the %s stream", wr ? "write" : "read",
                   what, name(st), x.wr ? "write" : "read", x.what, name(x.st));
          first = m;
        }
      }
    }
    log.push_back(Acc{what, a, b, st, vc[st][st], wr});
  }
  // Everything logged so far is ordered before the main stream's present: start the next call with an empty log.
  void new_call() { log.clear(); waits = 0; }
};

}  // namespace

struct PlbEngine {
  PlbConfig c;
  int64_t poff[PLB_NPARAM], psize[PLB_NPARAM], ptotal, ptrain;
  int E, H, I, L, NH, V, P, NP, NT;
  int64_t Tcap;   // padded token capacity
  int64_t NMcap;  // padded masked-row capacity
  // workspace offsets (bytes)
  int64_t o_wbf, o_wqkvT, o_wdT, o_w1T, o_w2T, o_wpT, o_winT;
  int64_t o_e, o_x, o_qkv, o_ctx, o_pre1, o_a, o_u, o_g, o_pre2;
  int64_t o_lse, o_delta, o_mean1, o_rstd1, o_mean2, o_rstd2;
  int64_t o_dqkv, o_dpre1, o_du, o_dpre2;
  int64_t o_dy0, o_dy1, o_da, o_dctx, o_de;
  int64_t o_hm, o_logm, o_dlog, o_dhm, o_rows, o_tgt, o_w, o_lrows;
  int64_t o_slab, o_part1, o_part2, o_parte, o_scratch, o_dxe, o_ducol, o_slab2, o_scratch2, o_qkvcol = 0;
  int qkvcol_rows = 0;  // partial rows per layer of the Q/K/V bias gradient: max_batch * ceil(max_seq / 128) * 4
  int64_t slab2_floats;
  // token (grapheme) head training: padded copies and the [Tp][NTp] logit / gradient images (NT > 0 only)
  int NTp = 0;
  int64_t o_bt = 0, o_wtT = 0, o_tdl = 0, o_tlrows = 0, o_tscr = 0, o_tgrad = 0, o_tloss = 0;
  int64_t o_tpmax = 0, o_tpsum = 0, o_ttl = 0, o_tlse = 0, o_tw = 0, o_ttgt = 0, o_tcolp = 0;
  bool tok_pad_zeroed = false;  // pad columns of the transposed copy are zeroed once
  bool tok_grads_live = false;  // the last loss call produced token-head gradients (AdamW then steps them)
  // side stream: the tail of the backward (embedding chain, bias / LayerNorm column sums) runs beside the
  // four large weight-gradient GEMMs
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int64_t slab_floats;
  int64_t ws_bytes;
  int ln_blocks, emb_blocks;
  int part_rows = 0;            // rows per layer reserved in o_part1 / o_part2
  int part_rows_used = 0;       // rows per layer the last backward wrote
  int ln_fuse = 3;              // bit 0: LayerNorm forward in the producing GEMM's epilogue, bit 1: LayerNorm backward
  int64_t o_lnx = 0, o_lnerr = 0, lnx_bytes = 0;
  // host-visible mirror of the hand-off error word (pinned, device-mapped): written by the last launch of every loss
  // call, read by plb_poll_status without synchronising
  unsigned int* host_err = nullptr;
  unsigned int* host_err_dev = nullptr;
  bool gelu_dstash_on = true;   // PLBERT_GELU_STASH=u restores the pre-activation stash
  bool u_is_derivative = false; // what the "u" slots hold after the last forward
  // fp8 mode (plb_set_fp8): transient 1-byte images of the fp8 GEMMs' activation / gradient operands, fp8 weight copies
  // and the per-(site, layer) delayed-scaling state [amax | scale | deq] (+ one entry per weight copy)
  bool fp8_on = false, fp8_ready = false, fp8_bwd_ready = false, fp8_wstale = true;
  bool fp8_tn = true;           // fp8 calls run the weight-gradient GEMMs on the 1-byte images too (PLBERT_FP8_TN=0: bf16 operands)
  bool tn8_call = false;        // ... decided per training call by its forward (shapes), read by its backward
  // per-layer 1-byte images [Ls][Tp][width] of every GEMM operand that is an activation (e4m3: layer input x, context,
  // attention-block output a, gelu output g) or a gradient (e5m2: dpre2, dU, dpre1, dQKV): read by the next NT GEMM and,
  // all layers at once, by the token-major weight-gradient GEMMs
  int64_t o_x8 = 0, o_a8 = 0, o_g8 = 0, o_c8 = 0, o_dp8 = 0, o_du8 = 0, o_dp18 = 0, o_dq8 = 0;
  int64_t o_wq8 = 0, o_wd8 = 0, o_w18 = 0, o_w28 = 0, o_w2T8 = 0, o_w1T8 = 0, o_wqT8 = 0, o_wdT8 = 0;
  int64_t o_f8amax = 0, o_f8scale = 0, o_f8deq = 0, o_f8stats = 0;
  int f8n = 0;
  bool infer = false;           // inference-only workspace: one layer of activations, no gradient stash
  // Last application on the masked rows only (a phoneme-only loss call: nothing but the masked positions' final hidden
  // states reaches the loss, so behind the attention of application L-1 only those rows are computed): decided by the
  // forward of a call, read by its backward. pruned_rows = the compact row count (a multiple of 128), 0 = the call was full.
  int pruned_rows = 0;
  int64_t last_call_rows[2] = {0, 0};   // token rows the last loss call ran the post-attention part of its last application on | of
  int tok_steps = 0;            // AdamW steps the token head has taken (its own bias correction)
  // data-parallel exchange (plb_comm_*): RCCL communicator, its stream, and the join event of the pieces in flight
  RcclComm comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_piece = nullptr, ev_comm_done = nullptr;
  bool overlap = true;          // issue the all-reduce piecewise inside plb_loss_fwd_bwd
  bool comm_pending = false;    // pieces were issued: plb_allreduce_grads / plb_adamw_step must join ev_comm_done
  bool grads_reduced = false;   // the gradients of the last loss call have been all-reduced
  int64_t piece_floats = 0;     // floats submitted as pieces by the current loss call (must add up to the gradient range)
  int32_t piece_count = 0;      // collectives the last step issued (pieces by the loss call + in-stream all-reduces)
  // the step's health word travels too (one float, summed over the ranks): every rank skips, or none
  hipEvent_t ev_status = nullptr;
  bool status_pending = false;  // the word's all-reduce is in flight on the communication stream
  int32_t status_collectives = 0;
  float* last_loss = nullptr;   // where the last loss call put its loss (plb_status_import turns it into NaN)
  HbAudit hb;
  // exchange trace (plb_comm_trace): timing events around every piece of the last loss call
  struct PieceTrace { int64_t a, b; hipEvent_t released, done; };
  bool trace_on = false;
  std::vector<PieceTrace> trace;
  std::vector<hipEvent_t> trace_pool;
  hipEvent_t tr_call0 = nullptr, tr_tail0 = nullptr, tr_tail1 = nullptr;
  bool tr_tail_valid = false;   // the last traced call reached its tail (a zero-loss call has none)
  // bound buffers
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
  char* ws = nullptr;

  template <typename T>
  T* at(int64_t off) const { return reinterpret_cast<T*>(ws + off); }
  bf16_t* wbf(int which) const { return at<bf16_t>(o_wbf) + poff[which]; }
  float* par(int which) const { return params + poff[which]; }
  float* grd(int which) const { return grads + poff[which]; }
};

extern "C" int plb_launch_gemm_nt_big(const PlbGemmNT* p, int tile, int act, int out_f32, hipStream_t stream);

static void layout_params(PlbEngine* e) {
  const int64_t V = e->V, E = e->E, H = e->H, I = e->I, P = e->P, NP = e->NP, NT = e->NT;
  int64_t sz[PLB_NPARAM];
  sz[PLB_WORD_EMB] = V * E; sz[PLB_POS_EMB] = P * E; sz[PLB_TYPE_EMB] = (int64_t)e->c.type_vocab_size * E;
  sz[PLB_EMB_LN_W] = E; sz[PLB_EMB_LN_B] = E;
  sz[PLB_MAP_W] = H * E; sz[PLB_MAP_B] = H;
  sz[PLB_LN2_W] = H; sz[PLB_LN2_B] = H;
  sz[PLB_Q_W] = sz[PLB_K_W] = sz[PLB_V_W] = H * H;
  sz[PLB_Q_B] = sz[PLB_K_B] = sz[PLB_V_B] = H;
  sz[PLB_DENSE_W] = H * H; sz[PLB_DENSE_B] = H;
  sz[PLB_LN1_W] = H; sz[PLB_LN1_B] = H;
  sz[PLB_FFN_W] = I * H; sz[PLB_FFN_B] = I;
  sz[PLB_FFNO_W] = H * I; sz[PLB_FFNO_B] = H;
  sz[PLB_HEAD_W] = NP * H; sz[PLB_HEAD_B] = NP;
  sz[PLB_POOL_W] = H * H; sz[PLB_POOL_B] = H;
  sz[PLB_TOK_W] = NT * H; sz[PLB_TOK_B] = NT;
  int64_t off = 0;
  for (int i = 0; i < PLB_NPARAM; ++i) {
    e->poff[i] = off;
    e->psize[i] = sz[i];
    off += sz[i];
    if (i == PLB_HEAD_B) e->ptrain = off;
  }
  e->ptotal = off;
}

// Row splits of a token-major weight-gradient GEMM. Shapes that fit the 256x256 pipeline kernel get
// one workgroup per CU (tiles x splits <= 256); the rest use the 128x128 kernel at ~3 workgroups per CU.
static bool tn_big(int64_t Mtot, int Ncols, int K) { return Ncols % 256 == 0 && K % 256 == 0 && Mtot >= 8192; }
// PLBERT_TN_SPLITS=xcd restores round 1's rule (8 * s splits, s * tiles <= 32: whole splits per XCD, but only 192-216
// of the 256 CUs busy on the model's shapes); default: as many splits as fit one workgroup per CU.
static bool tn_fill_chip() {
  static const bool v = [] { const char* e = getenv("PLBERT_TN_SPLITS"); return !(e && !strcmp(e, "xcd")); }();
  return v;
}
// PLBERT_TN_CUS = n (64..256, default 256): workgroups a big weight-gradient GEMM may occupy. The tail of the backward is
// where the gradient pieces travel; RCCL's kernels need CUs of their own and the one-workgroup-per-CU grids leave 4-16
// (dense.weight: 4). Lowering n trades GEMM width for CUs the collective finds free — a knob for the first real N > 1 run
// (bench.py reports the tail's GEMM time with and without the exchange), read once per process.
static int tn_cus() {
  static const int v = [] {
    const char* e = getenv("PLBERT_TN_CUS");
    const int n = e ? atoi(e) : 256;
    return (n >= 64 && n <= 256) ? n : 256;
  }();
  return v;
}
static int tn_splits(int64_t Mtot, int N, int K, int* rows_per_split) {
  const bool big = tn_big(Mtot, N, K);
  const int tiles = big ? (N / 256) * (K / 256) : ((N + 127) / 128) * ((K + 127) / 128);
  int splits = 768 / tiles;
  if (big) {
    // One workgroup per CU (128 KiB of LDS each): tiles * splits <= 256 and as close to it as the tile count allows —
    // 24 tiles (the two FFN weights) -> 10 splits = 240 workgroups, 27 (QKV) -> 9 = 243, 9 (dense) -> 28 = 252. The
    // kernel deals the (split, tile) pairs to the XCDs in contiguous runs (xcd_remap), so an XCD still streams a
    // contiguous range of token rows through its L2 (a split may straddle two XCDs).
    int s = 32 / tiles;
    if (s < 1) s = 1;
    splits = tiles >= 256 ? 1 : (tn_fill_chip() ? (tn_cus() / tiles > 0 ? tn_cus() / tiles : 1) : 8 * s);  // a wide output (token head) needs no row splits
  }
  const int64_t maxs = Mtot / 64;
  if (splits > maxs) splits = (int)maxs;
  if (splits < 1) splits = 1;
  int64_t rps = rup((Mtot + splits - 1) / splits, 64);
  splits = (int)((Mtot + rps - 1) / rps);
  *rows_per_split = (int)rps;
  return splits;
}

extern "C" int plb_create(const PlbConfig* cfg, PlbEngine** out) {
  if (!cfg || !out) return fail("plb_create: null argument");
  const PlbConfig& c = *cfg;
  if (c.hidden_size != c.num_attention_heads * 64) return fail("plb_create: head_dim must be 64 (hidden %d, heads %d)", c.hidden_size, c.num_attention_heads);
  if (c.embedding_size % 64 || c.embedding_size > 256) return fail("plb_create: embedding_size must be a multiple of 64, <= 256");
  if (c.hidden_size % 128 || c.hidden_size > 1024) return fail("plb_create: hidden_size must be a multiple of 128, <= 1024");
  if (c.intermediate_size % 128) return fail("plb_create: intermediate_size must be a multiple of 128");
  if (c.num_phonemes < 4 || c.num_phonemes % 4 || c.num_phonemes > 256) return fail("plb_create: num_phonemes must be a multiple of 4 in [4,256]");
  if (c.num_tokens < 0 || c.num_tokens % 4) return fail("plb_create: num_tokens must be a non-negative multiple of 4");
  if (c.max_seq < 1 || c.max_seq > c.max_position_embeddings) return fail("plb_create: max_seq must be in [1, max_position_embeddings]");
  if (c.max_batch < 1 || c.num_hidden_layers < 1 || c.vocab_size < 1 || c.type_vocab_size < 1) return fail("plb_create: bad sizes");
  PlbEngine* e = new (std::nothrow) PlbEngine();
  if (!e) return fail("plb_create: out of host memory");
  e->c = c;
  e->E = c.embedding_size; e->H = c.hidden_size; e->I = c.intermediate_size; e->L = c.num_hidden_layers;
  e->NH = c.num_attention_heads; e->V = c.vocab_size; e->P = c.max_position_embeddings;
  e->NP = c.num_phonemes; e->NT = c.num_tokens;
  layout_params(e);
  const int64_t E = e->E, H = e->H, I = e->I, L = e->L;
  const int64_t T = (int64_t)c.max_batch * c.max_seq;
  const int64_t Tp = rup(T, 128);
  e->infer = c.inference_only != 0;
  const bool tr = !e->infer;
  const int64_t Ls = tr ? L : 1;  // layers of activations kept
  e->Tcap = Tp;
  e->NMcap = Tp;
  // LayerNorm-backward partial rows (tools/ln_bench.py --blocks): 512 workgroups are 22 % faster than 1024 at H = 1024
  // (12.1 vs 15.5 us standalone) and level at H = 768 (2 % faster alone, +-0.01 ms inside the step, where they also halve
  // the side stream's pass over the partials); PLBERT_LN_BLOCKS overrides
  e->ln_blocks = 512;
  if (const char* v = getenv("PLBERT_LN_BLOCKS")) { const int n = atoi(v); if (n >= 64 && n <= 4096) e->ln_blocks = n; }
  e->emb_blocks = 2048;
  // LayerNorm in the GEMM epilogues (gemm_ln.hip): PLBERT_LN_FUSE = off | fwd | bwd | both (default both)
  if (const char* v = getenv("PLBERT_LN_FUSE"))
    e->ln_fuse = !strcmp(v, "off") ? 0 : !strcmp(v, "fwd") ? 1 : !strcmp(v, "bwd") ? 2 : 3;
  e->part_rows = e->ln_blocks > (int)(2 * Tp / 128) ? e->ln_blocks : (int)(2 * Tp / 128);
  if (const char* v = getenv("PLBERT_GELU_STASH")) e->gelu_dstash_on = strcmp(v, "u") != 0;
  if (const char* v = getenv("PLBERT_FP8_TN")) e->fp8_tn = strcmp(v, "0") != 0;
  if (const char* v = getenv("PLBERT_HB_AUDIT")) e->hb.on = strcmp(v, "0") != 0;
  Carve cv;
  // bf16 weight copies: the flat copy (+ slack so 128-row B tiles never leave the buffer) and transposes
  e->o_wbf = cv.take((e->ptotal + 256 * (H > I ? H : I)) * 2);
  if (tr) {
    e->o_wqkvT = cv.take(rup(H, 128) * 3 * H * 2);
    e->o_wdT = cv.take(rup(H, 128) * H * 2);
    e->o_w1T = cv.take(rup(H, 128) * I * 2);
    e->o_w2T = cv.take(rup(I, 128) * H * 2);
    e->o_wpT = cv.take(rup(H, 128) * 256 * 2);
    e->o_winT = cv.take(rup(E, 128) * H * 2);
  }
  // forward stash (training: every layer; inference: one layer, two ping-pong slots of x)
  e->o_e = cv.take(Tp * E * 2);
  e->o_x = cv.take((tr ? L + 1 : 2) * Tp * H * 2);
  e->o_qkv = cv.take(Ls * Tp * 3 * H * 2);
  e->o_ctx = cv.take(Ls * Tp * H * 2);
  e->o_pre1 = cv.take(Ls * Tp * H * 2);
  e->o_a = cv.take(Ls * Tp * H * 2);
  e->o_u = cv.take(Ls * Tp * I * 2);
  e->o_g = cv.take(Ls * Tp * I * 2);
  e->o_pre2 = cv.take(Ls * Tp * H * 2);
  const int64_t stat = (int64_t)c.max_batch * e->NH * c.max_seq * 4;
  e->o_lse = cv.take(Ls * stat);
  e->o_mean1 = cv.take(Ls * Tp * 4); e->o_rstd1 = cv.take(Ls * Tp * 4);
  e->o_mean2 = cv.take(Ls * Tp * 4); e->o_rstd2 = cv.take(Ls * Tp * 4);
  // exchange granules of the LayerNorm epilogues: [Tp/128][nbn][nbn][128][2] x 8 B, nbn <= 4 column tiles; zero between launches
  e->lnx_bytes = (Tp / 128) * 16 * 128 * 2 * 8;
  e->o_lnx = cv.take(e->lnx_bytes);
  e->o_lnerr = cv.take(256);
  if (tr) {
    e->o_delta = cv.take(stat);
    // backward stash (operands of the batched dW GEMMs)
    e->o_dqkv = cv.take(L * Tp * 3 * H * 2);
    e->o_dpre1 = cv.take(L * Tp * H * 2);
    e->o_du = cv.take(L * Tp * I * 2);
    e->o_dpre2 = cv.take(L * Tp * H * 2);
    e->o_dy0 = cv.take(Tp * H * 2); e->o_dy1 = cv.take(Tp * H * 2);
    e->o_da = cv.take(Tp * H * 2); e->o_dctx = cv.take(Tp * H * 2);
    e->o_de = cv.take(Tp * E * 2);
  }
  // loss rows
  const int64_t NM = e->NMcap;
  e->o_hm = cv.take(NM * H * 2);
  e->o_logm = cv.take(NM * 256 * 4);
  e->o_dlog = cv.take(NM * 256 * 2);
  e->o_rows = cv.take(NM * 4); e->o_tgt = cv.take(NM * 4); e->o_w = cv.take(NM * 4); e->o_lrows = cv.take(NM * 4);
  if (tr) {
    e->o_dhm = cv.take(NM * H * 2);
    // reductions
    int64_t slab = 0;
    {
      const int64_t Mtot = L * Tp;
      const int ntp = (int)rup(e->NT, 256);
      const int shapes[7][2] = {{(int)(3 * H), (int)H}, {(int)H, (int)H}, {(int)I, (int)H}, {(int)H, (int)I}, {(int)H, (int)E}, {e->NP, (int)H}, {ntp, (int)H}};
      for (int i = 0; i < (e->NT ? 7 : 6); ++i) {
        int rps;
        const int64_t mt = i < 4 ? Mtot : Tp;
        const int s = tn_splits(mt, shapes[i][0], shapes[i][1], &rps);
        const int64_t f = (int64_t)s * shapes[i][0] * shapes[i][1];
        if (f > slab) slab = f;
      }
    }
    e->slab_floats = slab;
    e->o_slab = cv.take(slab * 4);
    e->o_part1 = cv.take(L * e->part_rows * 3 * H * 4);  // per LN-backward block: dgamma | dbeta | column sums of dx
    e->o_part2 = cv.take(L * e->part_rows * 3 * H * 4);
    e->o_parte = cv.take((int64_t)e->emb_blocks * 2 * E * 4);
    e->o_dxe = cv.take(Tp * E * 4);
    e->o_ducol = cv.take(L * (2 * Tp / 128) * I * 4);  // column-sum partials of dU from the GEMM epilogue
    e->qkvcol_rows = c.max_batch * ((c.max_seq + 127) / 128) * 4;
    // ... of dQKV from the attention-backward stores, one set per application: the kernels only STORE their partial rows
    // (summing over the applications in place made every wave end on a global read-modify-write: +0.19 ms per step)
    e->o_qkvcol = cv.take((int64_t)L * e->qkvcol_rows * 3 * H * 4);
    e->o_scratch = cv.take(512 * (3 * H > I ? 3 * H : I) * 4);  // colsum partials: up to 512 row splits
    e->o_scratch2 = cv.take(512 * (3 * H > I ? 3 * H : I) * 4);
    {
      int rps;
      const int s2 = tn_splits(Tp, (int)H, (int)E, &rps);
      e->slab2_floats = (int64_t)s2 * H * E;
      e->o_slab2 = cv.take(e->slab2_floats * 4);
    }
  }
  if (e->NT) {
    const int64_t NTp = rup(e->NT, 256);
    e->NTp = (int)NTp;
    e->o_bt = cv.take(NTp * 4);
    e->o_tlrows = cv.take(Tp * 4);
    e->o_tpmax = cv.take(Tp * (NTp / 256) * 4);
    e->o_tpsum = cv.take(Tp * (NTp / 256) * 4);
    e->o_ttl = cv.take(Tp * 4); e->o_tlse = cv.take(Tp * 4); e->o_tw = cv.take(Tp * 4);
    e->o_ttgt = cv.take(Tp * 8);
    e->o_tloss = cv.take(256);
    // zero-padded [NTp,H] bf16 copy of the token head for the fused GEMM + CE passes (the flat copy's rows past NT
    // belong to the next tensor)
    if (tr) {
      e->o_wtT = cv.take(rup(H, 128) * NTp * 2);
      e->o_tdl = cv.take(Tp * NTp * 2);
      e->o_tscr = cv.take(32 * NTp * 4);
      e->o_tcolp = cv.take(2 * (Tp / 128) * NTp * 4);
      e->o_tgrad = cv.take(NTp * H * 4);
    }
  }
  // fp8 mode: 1-byte operand images of every layer kept (training: all of them, for the weight gradients) and the weight copies
  e->o_x8 = cv.take(Ls * Tp * H); e->o_a8 = cv.take(Ls * Tp * H); e->o_g8 = cv.take(Ls * Tp * I); e->o_c8 = cv.take(Ls * Tp * H);
  e->o_wq8 = cv.take(rup(3 * H, 128) * H + 256 * H);
  e->o_wd8 = cv.take(rup(H, 128) * H + 256 * H);
  e->o_w18 = cv.take(rup(I, 128) * H + 256 * H);
  e->o_w28 = cv.take(rup(H, 128) * I + 256 * I);
  if (tr) {
    e->o_dp8 = cv.take(L * Tp * H); e->o_du8 = cv.take(L * Tp * I); e->o_dp18 = cv.take(L * Tp * H); e->o_dq8 = cv.take(L * Tp * 3 * H);
    e->o_w2T8 = cv.take(rup(I, 128) * H + 256 * H);
    e->o_w1T8 = cv.take(rup(H, 128) * I + 256 * I);
    e->o_wqT8 = cv.take(rup(H, 128) * 3 * H + 256 * 3 * H);
    e->o_wdT8 = cv.take(rup(H, 128) * H + 256 * H);
  }
  e->f8n = 8 * (int)L + 8;
  e->o_f8amax = cv.take((int64_t)e->f8n * 64 * 16 * 4); e->o_f8scale = cv.take(e->f8n * 4); e->o_f8deq = cv.take(e->f8n * 4);
  e->o_f8stats = cv.take(8 * 8 * 4);   // per operand site: maxima history, clamped-call count, worst overshoot (rowops.hip: fp8_scales_kernel)
  e->ws_bytes = cv.off;
  *out = e;
  return 0;
}

extern "C" void plb_destroy(PlbEngine* e) {
  if (!e) return;
  (void)plb_comm_destroy(e);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->host_err) (void)hipHostFree(e->host_err);
  for (auto& t : e->trace) { (void)hipEventDestroy(t.released); (void)hipEventDestroy(t.done); }
  for (auto ev : e->trace_pool) (void)hipEventDestroy(ev);
  for (auto ev : {e->tr_call0, e->tr_tail0, e->tr_tail1}) if (ev) (void)hipEventDestroy(ev);
  delete e;
}

extern "C" int plb_param_layout(const PlbEngine* e, int64_t* offsets, int64_t* sizes, int64_t* total, int64_t* trainable) {
  if (!e) return fail("plb_param_layout: null engine");
  for (int i = 0; i < PLB_NPARAM; ++i) {
    if (offsets) offsets[i] = e->poff[i];
    if (sizes) sizes[i] = e->psize[i];
  }
  if (total) *total = e->ptotal;
  if (trainable) *trainable = e->ptrain;
  return 0;
}

extern "C" int64_t plb_workspace_bytes(const PlbEngine* e) { return e ? e->ws_bytes : -1; }

extern "C" int plb_bind(PlbEngine* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* workspace,
                        int64_t workspace_bytes) {
  if (!e || !params || !workspace) return fail("plb_bind: null argument");
  if (workspace_bytes < e->ws_bytes) return fail("plb_bind: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)e->ws_bytes);
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return fail("plb_bind: buffers must be 16-byte aligned");
  if ((uintptr_t)workspace & 255) return fail("plb_bind: workspace must be 256-byte aligned");
  e->params = params; e->grads = grads; e->m = exp_avg; e->v = exp_avg_sq;
  e->ws = (char*)workspace;
  if (!e->host_err) {  // once, outside any launch sequence
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && h) {
      void* d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
        e->host_err = (unsigned int*)h;
        e->host_err_dev = (unsigned int*)d;
        *e->host_err = 0;
      } else {
        (void)hipHostFree(h);
      }
    }
    if (!e->host_err) return fail("plb_bind: cannot allocate the pinned status word");
  }
  if (!e->side && grads && !getenv("PLBERT_NO_SIDE_STREAM")) {  // created once, outside any launch sequence (a step may be graph-captured)
    // (default priority: at the highest one the step measured the same, 10.20-10.22 ms either way)
    if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) e->side = nullptr;
    if (e->side && (hipEventCreateWithFlags(&e->ev_fork, kStreamOrderEvent) != hipSuccess ||
                    hipEventCreateWithFlags(&e->ev_join, kStreamOrderEvent) != hipSuccess)) {
      (void)hipStreamDestroy(e->side);
      e->side = nullptr;
    }
  }
  return 0;
}

#define TRY(x)                                                                    \
  do {                                                                            \
    int rc_ = (x);                                                                \
    if (rc_) return fail("%s failed (rc %d) at %s:%d", #x, rc_, __FILE__, __LINE__); \
  } while (0)
#define HIPTRY(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return fail("%s: %s at %s:%d", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Stream-ordering calls of the engine go through these two: the HIP call, and — audit on — the same step in the model.
static int hb_idx(const PlbEngine* e, hipStream_t s) {
  if (e->side && s == e->side) return HbAudit::SIDE;
  if (e->comm_stream && s == e->comm_stream) return HbAudit::COMM;
  return HbAudit::MAIN;
}
static hipError_t ev_record(PlbEngine* e, hipEvent_t ev, hipStream_t s) {
  const hipError_t r = hipEventRecord(ev, s);
  if (e->hb.on) e->hb.record(ev, hb_idx(e, s));
  return r;
}
static hipError_t ev_wait(PlbEngine* e, hipStream_t s, hipEvent_t ev) {
  const hipError_t r = hipStreamWaitEvent(s, ev, 0);
  if (e->hb.on) e->hb.wait(hb_idx(e, s), ev);
  return r;
}
// a launch (or memset / collective) enqueued on `stream` reads / writes these bytes
#define HB_R(stream, ptr, bytes, what) do { if (e->hb.on) e->hb.access(hb_idx(e, stream), (ptr), (size_t)(bytes), false, what); } while (0)
#define HB_W(stream, ptr, bytes, what) do { if (e->hb.on) e->hb.access(hb_idx(e, stream), (ptr), (size_t)(bytes), true, what); } while (0)

// ---- fp8 mode ---------------------------------------------------------------------------------------------------------
// Sites: activations X (layer input), A (attention block output), G (gelu output) in e4m3; gradients DP (dpre2) and DU
// in e5m2 (their range within a tensor is what e5m2's five exponent bits are for); weights W* in e4m3.
enum { F8_X = 0, F8_A, F8_G, F8_C, F8_DP, F8_DU, F8_DP1, F8_DQ, F8_NSITE };   // 4 activation sites, then 4 gradient sites
enum { F8_AMAX_WORDS = 64 * 16 };  // floats per site in the amax array (common.h: F8_SLOTS x F8_STRIDE)
enum { F8W_QKV = 0, F8W_D, F8W_1, F8W_2, F8W_2T, F8W_1T, F8W_QKVT, F8W_DT, F8W_N };
static int f8_site(const PlbEngine* e, int site, int l) { return site * e->L + l; }
static int f8_w(const PlbEngine* e, int w) { return F8_NSITE * e->L + w; }
static float* f8_amax(const PlbEngine* e, int i) { return e->at<float>(e->o_f8amax) + (int64_t)i * F8_AMAX_WORDS; }
static float* f8_scale(const PlbEngine* e, int i) { return e->at<float>(e->o_f8scale) + i; }
static float* f8_deq(const PlbEngine* e, int i) { return e->at<float>(e->o_f8deq) + i; }
// every GEMM of the fp8 set has a pipeline-tile form at this token count (else the whole call runs in bf16)
static bool fp8_shapes_ok(const PlbEngine* e, int64_t Tp) {
  const int64_t H = e->H, I = e->I;
  if (!(H == 768 || H == 1024) || I % 256 || H % 128 || I % 128) return false;
  if (Tp % 128) return false;
  return (3 * H) % 384 == 0 || (3 * H) % 256 == 0;
}
// per-tensor e4m3 copies of the fp8 GEMMs' weights.
// exact = true (after plb_sync_weights / plb_set_fp8: the weights may be anything): maximum, scale, quantisation — three
//   passes. exact = false (after an AdamW step): ONE launch quantises all copies with the scale the previous
//   quantisation's maxima give and records the new maxima (a weight moves by <= lr per step; values are clamped).
static int fp8_quantize_weights(PlbEngine* e, hipStream_t s, bool exact = true) {
  const int H = e->H, I = e->I;
  struct W { int w; const void* src; int bf16; int rows, cols; int64_t dst; } ws[F8W_N] = {
      {F8W_QKV, e->par(PLB_Q_W), 0, 3 * H, H, e->o_wq8},
      {F8W_D, e->par(PLB_DENSE_W), 0, H, H, e->o_wd8},
      {F8W_1, e->par(PLB_FFN_W), 0, I, H, e->o_w18},
      {F8W_2, e->par(PLB_FFNO_W), 0, H, I, e->o_w28},
      {F8W_2T, e->infer ? nullptr : e->at<bf16_t>(e->o_w2T), 1, I, H, e->o_w2T8},
      {F8W_1T, e->infer ? nullptr : e->at<bf16_t>(e->o_w1T), 1, H, I, e->o_w1T8},
      {F8W_QKVT, e->infer ? nullptr : e->at<bf16_t>(e->o_wqkvT), 1, H, 3 * H, e->o_wqT8},
      {F8W_DT, e->infer ? nullptr : e->at<bf16_t>(e->o_wdT), 1, H, H, e->o_wdT8}};
  if (exact) {
    HIPTRY(hipMemsetAsync(f8_amax(e, f8_w(e, 0)), 0, F8W_N * F8_AMAX_WORDS * sizeof(float), s));
    for (auto& w : ws) {
      if (!w.src) continue;
      TRY(plb_launch_amax(w.src, w.bf16, (size_t)w.rows, w.cols, w.cols, f8_amax(e, f8_w(e, w.w)), s));
    }
  }
  // amax -> scale (and the maxima are cleared: the quantisation below records this step's)
  TRY(plb_launch_fp8_scales(f8_amax(e, f8_w(e, 0)), f8_scale(e, f8_w(e, 0)), f8_deq(e, f8_w(e, 0)), F8W_N, 448.f, 1, s));
  const void* src[8]; int bf[8]; size_t n[8]; const float* sc[8]; uint8_t* dst[8]; float* am[8];
  int k = 0;
  for (auto& w : ws) {
    if (!w.src) continue;
    src[k] = w.src; bf[k] = w.bf16; n[k] = (size_t)w.rows * w.cols; sc[k] = f8_scale(e, f8_w(e, w.w));
    dst[k] = e->at<uint8_t>(w.dst); am[k] = f8_amax(e, f8_w(e, w.w));
    ++k;
  }
  TRY(plb_launch_quantize_multi(k, src, bf, n, sc, dst, am, s));
  e->fp8_wstale = false;
  return 0;
}
// end of a call in fp8 mode: this call's maxima become the next call's scales (delayed scaling, history 1)
static int fp8_update_scales(PlbEngine* e, hipStream_t s) {
  const int L = e->L;
  // One scale per SITE, shared by its L applications (their maxima are recorded per application): the weight-gradient
  // GEMMs sum the products of two images over all applications under one dequantisation factor.
  // X, A, G, C: e4m3, 448. Gradients (DP, DU, DP1, DQ): e5m2, mapped to HALF the format's range — a step whose gradients
  // are up to 2x the previous step's (a smaller batch: the loss is a mean over samples) still fits; five exponent bits
  // have the binade to spare. One launch for the whole site table.
  // Every site: the scale comes from the LARGEST maximum of the last four calls (a call whose gradients are a multiple of the
  // previous call's — or whose batch simply has larger activations than the previous one: alternating batches clamped the
  // gelu site in a third of the calls of a 20,000-step soak under a history of one — is clamped only beyond that), and every
  // site counts the calls in which values were clamped (plb_fp8_stats): a clamped step is visible instead of silent.
  TRY(plb_launch_fp8_scales2(f8_amax(e, 0), f8_scale(e, 0), f8_deq(e, 0), 8 * L, 448.f, L, 4 * L, 28672.f,
                             e->at<float>(e->o_f8stats), 0, s));
  return 0;
}

static int sync_transposes(PlbEngine* e, hipStream_t s, bool exact_fp8 = true) {
  const int H = e->H, I = e->I, E = e->E;

  if (e->NT) {  // bias of the token head padded to NTp columns (fused GEMM + CE passes)
    if (!e->tok_pad_zeroed) HIPTRY(hipMemsetAsync(e->at<float>(e->o_bt), 0, (size_t)e->NTp * 4, s));
    HIPTRY(hipMemcpyAsync(e->at<float>(e->o_bt), e->par(PLB_TOK_B), (size_t)e->NT * 4, hipMemcpyDeviceToDevice, s));
  }
  if (e->infer) {  // the transposed copies serve the backward only
    e->tok_pad_zeroed = true;
    if (e->fp8_on) return fp8_quantize_weights(e, s, exact_fp8 || e->fp8_wstale);
    return 0;
  }
  // fused QKV [3H,H] -> [H,3H]; the three tensors are adjacent in the flat buffer
  // (one launch for all of them: each is a few microseconds of work behind 5 us of launch latency)
  const float* tsrc[8] = {e->par(PLB_Q_W), e->par(PLB_DENSE_W), e->par(PLB_FFN_W), e->par(PLB_FFNO_W), e->par(PLB_HEAD_W),
                          e->par(PLB_MAP_W)};
  bf16_t* tdst[8] = {e->at<bf16_t>(e->o_wqkvT), e->at<bf16_t>(e->o_wdT), e->at<bf16_t>(e->o_w1T), e->at<bf16_t>(e->o_w2T),
                     e->at<bf16_t>(e->o_wpT), e->at<bf16_t>(e->o_winT)};
  int tR[8] = {3 * H, H, I, H, e->NP, H}, tC[8] = {H, H, H, I, H, E}, tld[8] = {3 * H, H, I, H, 256, H};
  int nt = 6;
  if (e->NT) {  // transposed weight of the token head, padded to NTp columns
    if (!e->tok_pad_zeroed) HIPTRY(hipMemsetAsync(e->at<bf16_t>(e->o_wtT), 0, (size_t)rup(H, 128) * e->NTp * 2, s));
    tsrc[nt] = e->par(PLB_TOK_W); tdst[nt] = e->at<bf16_t>(e->o_wtT); tR[nt] = e->NT; tC[nt] = H; tld[nt] = e->NTp;
    ++nt;
  }
  TRY(plb_launch_transpose_cast_multi(nt, tsrc, tR, tC, tdst, tld, s));
  e->tok_pad_zeroed = true;
  if (e->fp8_on) return fp8_quantize_weights(e, s, exact_fp8 || e->fp8_wstale);
  return 0;
}

extern "C" int plb_set_fp8(PlbEngine* e, int32_t on, void* stream) {
  if (!e || !e->ws) return fail("plb_set_fp8: engine not bound");
  if (on && !(e->H == 768 || e->H == 1024)) return fail("plb_set_fp8: the fp8 path needs hidden_size 768 or 1024");
  if (on && !e->fp8_on) {  // the first call afterwards runs in bf16 and calibrates the scales
    hipStream_t s = (hipStream_t)stream;
    HIPTRY(hipMemsetAsync(f8_amax(e, 0), 0, (size_t)e->f8n * F8_AMAX_WORDS * 4, s));
    HIPTRY(hipMemsetAsync(e->at<float>(e->o_f8stats), 0, 8 * 8 * 4, s));
    HIPTRY(hipMemsetAsync(f8_scale(e, 0), 0, (size_t)e->f8n * 4, s));   // "no scale yet": the calibration call's maxima are not overshoots
    e->fp8_ready = false;      // activation sites: armed by the first forward
    e->fp8_bwd_ready = false;  // gradient sites: armed by the first backward
    e->fp8_wstale = true;
  }
  e->fp8_on = on != 0;
  return 0;
}
extern "C" int plb_fp8_state(const PlbEngine* e, int32_t* enabled, int32_t* calibrated) {
  if (!e) return fail("plb_fp8_state: null engine");
  if (enabled) *enabled = e->fp8_on;
  if (calibrated) *calibrated = e->fp8_ready;
  return 0;
}

// Per operand site (X, A, G, C in e4m3; dpre2, dU, dpre1, dQKV in e5m2): calls since plb_set_fp8 in which the site's values
// exceeded the format's range under the delayed scale they were quantised with (those elements were clamped), and the
// worst overshoot (true maximum x scale / format maximum; <= 1 = never clamped). Synchronises `stream`.
extern "C" int plb_fp8_stats(PlbEngine* e, float clamped_calls[8], float worst_overshoot[8], void* stream) {
  if (!e || !e->ws) return fail("plb_fp8_stats: engine not bound");
  float st[64];
  HIPTRY(hipMemcpyAsync(st, e->at<float>(e->o_f8stats), sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPTRY(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < 8; ++i) {
    if (clamped_calls) clamped_calls[i] = st[i * 8 + 4];
    if (worst_overshoot) worst_overshoot[i] = st[i * 8 + 5];
  }
  return 0;
}

extern "C" int plb_sync_weights(PlbEngine* e, void* stream) {
  if (!e || !e->ws) return fail("plb_sync_weights: engine not bound");
  hipStream_t s = (hipStream_t)stream;
  TRY(plb_launch_cast_bf16(e->params, e->at<bf16_t>(e->o_wbf), (size_t)e->ptotal, s));
  return sync_transposes(e, s);
}

static int check_shape(const PlbEngine* e, int B, int S, const char* who) {
  if (!e || !e->ws) return fail("%s: engine not bound", who);
  if (B < 1 || S < 1 || B > e->c.max_batch || S > e->c.max_seq || (int64_t)B * S > (int64_t)e->c.max_batch * e->c.max_seq)
    return fail("%s: batch %d x seq %d exceeds the engine capacity %d x %d", who, B, S, e->c.max_batch, e->c.max_seq);
  return 0;
}

// One NT GEMM of the fp8 set: the fp8 launch when the call runs in fp8 mode (A8 / B8 images, their dequantisation
// factors), else the bf16 launch on A / B. g carries everything else (shapes, bias, residual, outputs).
struct F8Op { const uint8_t* A8; const uint8_t* B8; const float* deq_a; const float* deq_b; int a_bf8; };
static PlbGemmNT f8_operands(const PlbGemmNT* g, const F8Op* f8) {
  PlbGemmNT q = *g;
  q.A = reinterpret_cast<const bf16_t*>(f8->A8); q.B = reinterpret_cast<const bf16_t*>(f8->B8);
  q.deq_a = f8->deq_a; q.deq_b = f8->deq_b;
  return q;
}
static int gemm_nt_any(PlbGemmNT* g, int act, const F8Op* f8, hipStream_t s) {
  if (!f8) return plb_launch_gemm_nt(g, act, 0, s);
  PlbGemmNT q = f8_operands(g, f8);
  return plb_launch_gemm_nt_fp8(&q, act, f8->a_bf8, s);
}
// the LayerNorm forms (5 / 6) and the gelu-derivative-stash forms, bf16 or fp8 operands
static int gemm_nt_ln_any(PlbGemmNT* g, int mode, const F8Op* f8, hipStream_t s) {
  if (!f8) return plb_launch_gemm_nt_ln(g, mode, s);
  PlbGemmNT q = f8_operands(g, f8);
  return plb_launch_gemm_nt_fp8_ln(&q, mode, f8->a_bf8, s);
}
static int gemm_nt_gelud_any(PlbGemmNT* g, int backward, const F8Op* f8, hipStream_t s) {
  if (!f8) return plb_launch_gemm_nt_gelud(g, backward, s);
  PlbGemmNT q = f8_operands(g, f8);
  return plb_launch_gemm_nt_fp8_gelud(&q, backward, f8->a_bf8, s);
}
// the 1-byte image + running maximum a launch writes beside its output (site i: scale in, maxima out)
static void f8_out(const PlbEngine* e, PlbGemmNT* g, uint8_t* img, int ld, int site, int bf8) {
  g->C8 = img; g->ldc8 = ld; g->q_scale = f8_scale(e, site); g->q_amax = f8_amax(e, site); g->c8_bf8 = bf8;
}

// LayerNorm in the epilogue of the GEMM that produces its input (gemm_ln.hip, gemm_fp8_ln.hip): the shapes it exists for,
// in bf16 and in fp8 calls alike (the fp8 forms write the 1-byte images the standalone LayerNorm kernels used to write);
// an fp8 CALIBRATION call computes in bf16 (it must equal the bf16 path bit for bit: tests/test_gpu_fp8.py).
static bool ln_fusable(const PlbEngine* e, int64_t Tp, int bit) {
  const int H = e->H;
  return (e->ln_fuse & bit) && Tp % 1024 == 0 && (H % 384 == 0 ? H / 384 : H % 256 == 0 ? H / 256 : 99) <= 4;
}
// Does the forward of this call stash gelu_new'(u) instead of u (plb_launch_gemm_nt_gelud)? Recorded in the engine: the
// backward of the same call must read the stash the way the forward wrote it — the stash is in the LANE layout of the
// tile that wrote it (256x256 in bf16 calls, 128x256 in fp8 calls), so forward and backward of a call run in one mode.
static bool gelu_dstash(PlbEngine* e, int64_t Tp, bool f8_call) {
  e->u_is_derivative = e->gelu_dstash_on && (f8_call ? Tp % 128 == 0 : Tp % 256 == 0) && e->I % 256 == 0;
  return e->u_is_derivative;
}
static void ln_fields(const PlbEngine* e, PlbGemmNT* g, const float* gamma, const float* beta, float* mean, float* rstd) {
  g->ln_gamma = gamma; g->ln_beta = beta; g->ln_mean = mean; g->ln_rstd = rstd; g->ln_eps = e->c.layer_norm_eps;
  g->ln_xchg = e->at<unsigned long long>(e->o_lnx); g->ln_err = e->at<unsigned int>(e->o_lnerr);
}

// Embeddings + L applications of the shared layer. stash: keep every layer's activations (training)
// or reuse the layer-0 slots (inference). Returns the final hidden buffer in *xout.
// fp8 mode: EVERY large GEMM of the layer runs on 1-byte images — QKV, dense (+ LayerNorm 1), FFN up (+ gelu), FFN output
// (+ LayerNorm 2) on e4m3 images of x, the attention context, a and gelu(u). Each image is written, with the scale its
// site learnt in the previous call, by the launch that produces the tensor (the fused LayerNorm / gelu epilogues, the
// attention kernel; the standalone LayerNorm kernels on shapes without a fused form), one image per layer in a training
// call: the weight-gradient GEMMs read them all at the end of the backward. A calibration call (the first after
// plb_set_fp8, and a training call whose gradient sites have not been seen yet) runs in bf16 and only records the maxima.
static bool tn8_ok(const PlbEngine* e, int64_t Mtot);
static bool f8_call(const PlbEngine* e, int64_t Tp, bool train) {
  return e->fp8_on && e->fp8_ready && (!train || e->fp8_bwd_ready) && fp8_shapes_ok(e, Tp);
}
// ---- the last application on the masked rows only --------------------------------------------------------------------
// The reference evaluates every position of every application and then reads the masked positions of the LAST one
// (train.py:107-131: pred[b, :len_b][idx_b]). Positions exchange information only inside attention (keys / values), so
// behind the attention of application L-1 nothing a non-masked row computes reaches the loss — forward or backward, where
// its output gradient is exactly zero. A phoneme-only loss call therefore runs dense + LayerNorm, the FFN and the second
// LayerNorm of application L-1 on the ~13 % masked rows alone (gathered, padded to 128), and their backward likewise; Q/K/V
// projection and attention stay on all rows (every key / value is needed), and so does everything below application L-1.
// Results are those of the full evaluation (each row's arithmetic is unchanged; the weight gradients lose only exact
// zeros from their sums). Compact GEMMs of ~2,300 rows do not fill one-tile-per-CU grids, so this part runs on the
// small-shape launches (GEMM + LayerNorm kernels, gelu by act 1 / 2): 168 -> 75 us forward, 164 -> 86 us backward, and the
// three weight-gradient GEMMs that stack its rows read (L-1) Tp + Mc rows instead of L Tp (measured: profiles/r05_*).
// fp8 calls run this part in bf16 too and add the 1-byte images of the compact rows that their stacked weight-gradient
// GEMMs read. Not taken by dual-head calls (the token loss reads every position), when more than half of the positions
// are masked, and under PLBERT_PRUNE_LAST=0.
struct Prune { const int32_t* rows; int n; int Mc; };
static int g_prune_last = -1;   // test / tuning hook (plb_set_prune_last): -1 the environment's choice, 0 off, 1 on
extern "C" void plb_set_prune_last(int on) { g_prune_last = on < 0 ? -1 : (on ? 1 : 0); }
static bool prune_enabled() {
  static const bool v = [] { const char* e = getenv("PLBERT_PRUNE_LAST"); return !(e && !strcmp(e, "0")); }();
  return g_prune_last < 0 ? v : g_prune_last != 0;
}
// post-attention part of application L-1 on the compact rows; leaves the final hidden rows in o_hm ([Mc][H]: the head's
// operand) and, in a training call, the compact activations at the START of application L-1's stash slots — the stacked
// weight-gradient operands then simply end Tp - Mc rows earlier
static int last_application_fwd_pruned(PlbEngine* e, const Prune* pr, bool stash, bool calib, bool tn8, const bf16_t* x,
                                       const bf16_t* ctx_att, int64_t Tp, hipStream_t s) {
  const int H = e->H, I = e->I, L = e->L, Mc = pr->Mc, n = pr->n;
  const int64_t sl = stash ? L - 1 : 0;
  const int64_t Tcap = Tp;   // slots of a call are packed with the call's own padded token count
  bf16_t* ctx_s = e->at<bf16_t>(e->o_ctx) + sl * Tcap * H;
  bf16_t* pre1_s = e->at<bf16_t>(e->o_pre1) + sl * Tcap * H;
  bf16_t* a_s = e->at<bf16_t>(e->o_a) + sl * Tcap * H;
  bf16_t* u = e->at<bf16_t>(e->o_u) + sl * Tcap * I;
  bf16_t* gl = e->at<bf16_t>(e->o_g) + sl * Tcap * I;
  bf16_t* pre2 = e->at<bf16_t>(e->o_pre2) + sl * Tcap * H;
  // training: compact tensors in their own slots (the backward and the weight gradients read them), the gathered
  // residual rows in a backward temporary; forward-only: the one set of slots, rotated so that nothing is read and
  // written by the same launch (ctx_att = the ctx slot: gathered into the pre1 slot, whose sum then goes to the ctx slot)
  bf16_t* ctxc = stash ? ctx_s : pre1_s;
  bf16_t* xc = stash ? e->at<bf16_t>(e->o_da) : a_s;
  bf16_t* pre1c = stash ? pre1_s : ctx_s;
  bf16_t* ac = stash ? a_s : pre1_s;
  bf16_t* hm = e->at<bf16_t>(e->o_hm);
  TRY(plb_launch_gather_rows(ctx_att, H, pr->rows, n, Mc, H, ctxc, H, s));
  TRY(plb_launch_gather_rows(x, H, pr->rows, n, Mc, H, xc, H, s));
  PlbGemmNT g;
  memset(&g, 0, sizeof(g));
  g.A = ctxc; g.lda = H; g.B = e->wbf(PLB_DENSE_W); g.ldb = H; g.M = Mc; g.N = H; g.K = H; g.Mstore = Mc;
  g.bias = e->par(PLB_DENSE_B); g.res = xc; g.ldr = H; g.C = pre1c; g.ldc = H;
  TRY(plb_launch_gemm_nt(&g, 0, 0, s));
  PlbLayerNorm ln;
  memset(&ln, 0, sizeof(ln));
  ln.x = pre1c; ln.ldx = H; ln.gamma = e->par(PLB_LN1_W); ln.beta = e->par(PLB_LN1_B); ln.eps = e->c.layer_norm_eps;
  ln.y = ac; ln.ldy = H; ln.T = Mc; ln.H = H;
  ln.mean = e->at<float>(e->o_mean1) + sl * Tcap; ln.rstd = e->at<float>(e->o_rstd1) + sl * Tcap;
  TRY(plb_launch_ln_fwd(&ln, s));
  if (calib) TRY(plb_launch_amax(ac, 1, (size_t)n, H, H, f8_amax(e, f8_site(e, F8_A, L - 1)), s));
  memset(&g, 0, sizeof(g));
  g.A = ac; g.lda = H; g.B = e->wbf(PLB_FFN_W); g.ldb = H; g.M = Mc; g.N = I; g.K = H; g.Mstore = Mc;
  g.bias = e->par(PLB_FFN_B); g.C = u; g.ldc = I; g.C2 = gl; g.ldc2 = I;
  TRY(plb_launch_gemm_nt(&g, 1, 0, s));
  if (calib) TRY(plb_launch_amax(gl, 1, (size_t)n, I, I, f8_amax(e, f8_site(e, F8_G, L - 1)), s));
  memset(&g, 0, sizeof(g));
  g.A = gl; g.lda = I; g.B = e->wbf(PLB_FFNO_W); g.ldb = I; g.M = Mc; g.N = H; g.K = I; g.Mstore = Mc;
  g.bias = e->par(PLB_FFNO_B); g.res = ac; g.ldr = H; g.C = pre2; g.ldc = H;
  TRY(plb_launch_gemm_nt(&g, 0, 0, s));
  memset(&ln, 0, sizeof(ln));
  ln.x = pre2; ln.ldx = H; ln.gamma = e->par(PLB_LN2_W); ln.beta = e->par(PLB_LN2_B); ln.eps = e->c.layer_norm_eps;
  ln.y = hm; ln.ldy = H; ln.T = Mc; ln.H = H;
  ln.mean = e->at<float>(e->o_mean2) + sl * Tcap; ln.rstd = e->at<float>(e->o_rstd2) + sl * Tcap;
  TRY(plb_launch_ln_fwd(&ln, s));
  if (tn8) {
    // fp8 training call: this part itself runs in bf16 (2,000 rows: nothing to gain from fp8 operands), but the stacked
    // weight-gradient GEMMs read 1-byte images of EVERY application: the compact context / a / gelu(u) rows as e4m3 images
    // at the start of this application's image slots, under the sites' scales, their maxima reported like any other's
    const void* src[3] = {ctxc, ac, gl}; const int fl[3] = {1, 1, 1};
    const size_t nel[3] = {(size_t)Mc * H, (size_t)Mc * H, (size_t)Mc * I};
    const float* sc[3] = {f8_scale(e, f8_site(e, F8_C, L - 1)), f8_scale(e, f8_site(e, F8_A, L - 1)), f8_scale(e, f8_site(e, F8_G, L - 1))};
    uint8_t* dst[3] = {e->at<uint8_t>(e->o_c8) + sl * Tcap * H, e->at<uint8_t>(e->o_a8) + sl * Tcap * H, e->at<uint8_t>(e->o_g8) + sl * Tcap * I};
    float* am[3] = {f8_amax(e, f8_site(e, F8_C, L - 1)), f8_amax(e, f8_site(e, F8_A, L - 1)), f8_amax(e, f8_site(e, F8_G, L - 1))};
    TRY(plb_launch_quantize_multi(3, src, fl, nel, sc, dst, am, s));
  }
  return 0;
}

static int run_encoder(PlbEngine* e, const int64_t* ids, const int32_t* lengths, int B, int S, bool stash, bf16_t** xout,
                       hipStream_t s, const Prune* pr = nullptr) {
  const int E = e->E, H = e->H, I = e->I, L = e->L;
  const int T = B * S;
  const int64_t Tp = rup(T, 128);
  const bool f8 = f8_call(e, Tp, stash);
  const bool calib = e->fp8_on && !f8;
  if (e->fp8_on && e->fp8_wstale) TRY(fp8_quantize_weights(e, s));
  PlbEmbed em;
  memset(&em, 0, sizeof(em));
  em.ids = ids; em.T = T; em.S = S; em.E = E; em.V = e->V;
  em.word = e->par(PLB_WORD_EMB); em.pos = e->par(PLB_POS_EMB); em.type0 = e->par(PLB_TYPE_EMB);
  em.gamma = e->par(PLB_EMB_LN_W); em.beta = e->par(PLB_EMB_LN_B); em.eps = e->c.layer_norm_eps;
  em.out = e->at<bf16_t>(e->o_e); em.ldo = E;
  TRY(plb_launch_embed_fwd(&em, s));

  bf16_t* xall = e->at<bf16_t>(e->o_x);
  PlbGemmNT g;
  memset(&g, 0, sizeof(g));
  g.A = em.out; g.lda = E; g.B = e->wbf(PLB_MAP_W); g.ldb = E; g.M = (int)Tp; g.N = H; g.K = E; g.Mstore = (int)Tp;
  g.bias = e->par(PLB_MAP_B); g.C = xall; g.ldc = H;
  TRY(plb_launch_gemm_nt(&g, 0, 0, s));
  if (f8) {  // layer 0 reads the map-in output, which no LayerNorm produced: one quantisation pass
    // (image and the site's maximum in one pass: the rows are contiguous)
    const void* src1[1] = {xall}; const int bf1[1] = {1}; const size_t n1[1] = {(size_t)T * H};
    const float* sc1[1] = {f8_scale(e, f8_site(e, F8_X, 0))}; uint8_t* dst1[1] = {e->at<uint8_t>(e->o_x8)};
    float* am1[1] = {f8_amax(e, f8_site(e, F8_X, 0))};
    TRY(plb_launch_quantize_multi(1, src1, bf1, n1, sc1, dst1, am1, s));
  }
  const bool fuse_f = ln_fusable(e, Tp, 1);
  const bool dstash = gelu_dstash(e, Tp, f8);
  // do the weight-gradient GEMMs of this call read the 1-byte images? Then gelu(u), dU and dQKV leave as images alone.
  // (Needs the derivative stash: forms 1 / 2 always write their bf16 outputs.)
  if (stash) e->tn8_call = f8 && e->fp8_tn && dstash && tn8_ok(e, (int64_t)L * Tp);
  const bool tn8 = stash && e->tn8_call;

  for (int l = 0; l < L; ++l) {
    const int64_t sl = stash ? l : 0;
    bf16_t* x = stash ? xall + (int64_t)l * Tp * H : xall + (int64_t)(l & 1) * Tp * H;
    bf16_t* y = stash ? xall + (int64_t)(l + 1) * Tp * H : xall + (int64_t)((l + 1) & 1) * Tp * H;
    bf16_t* qkv = e->at<bf16_t>(e->o_qkv) + sl * Tp * 3 * H;
    bf16_t* ctx = e->at<bf16_t>(e->o_ctx) + sl * Tp * H;
    bf16_t* pre1 = e->at<bf16_t>(e->o_pre1) + sl * Tp * H;
    bf16_t* a = e->at<bf16_t>(e->o_a) + sl * Tp * H;
    bf16_t* u = e->at<bf16_t>(e->o_u) + sl * Tp * I;
    bf16_t* gl = e->at<bf16_t>(e->o_g) + sl * Tp * I;
    bf16_t* pre2 = e->at<bf16_t>(e->o_pre2) + sl * Tp * H;
    // this layer's 1-byte images; the next layer's input image (inference: the one slot, consumed before it is rewritten)
    uint8_t* x8 = e->at<uint8_t>(e->o_x8) + sl * Tp * H;
    uint8_t* x8n = e->at<uint8_t>(e->o_x8) + (stash ? (int64_t)(l + 1) : 0) * Tp * H;
    uint8_t* c8 = e->at<uint8_t>(e->o_c8) + sl * Tp * H;
    uint8_t* a8 = e->at<uint8_t>(e->o_a8) + sl * Tp * H;
    uint8_t* g8 = e->at<uint8_t>(e->o_g8) + sl * Tp * I;
    const int sX = f8_site(e, F8_X, l), sA = f8_site(e, F8_A, l), sG = f8_site(e, F8_G, l), sC = f8_site(e, F8_C, l);
    if (calib) TRY(plb_launch_amax(x, 1, (size_t)T, H, H, f8_amax(e, sX), s));
    // fused QKV projection
    memset(&g, 0, sizeof(g));
    g.A = x; g.lda = H; g.B = e->wbf(PLB_Q_W); g.ldb = H; g.M = (int)Tp; g.N = 3 * H; g.K = H; g.Mstore = (int)Tp;
    g.bias = e->par(PLB_Q_B); g.C = qkv; g.ldc = 3 * H;
    F8Op oq = {x8, e->at<uint8_t>(e->o_wq8), f8_deq(e, sX), f8_deq(e, f8_w(e, F8W_QKV)), 0};
    TRY(gemm_nt_any(&g, 0, f8 ? &oq : nullptr, s));
    PlbAttn at;
    memset(&at, 0, sizeof(at));
    at.qkv = qkv; at.ldqkv = 3 * H; at.lengths = lengths; at.B = B; at.S = S; at.NH = e->NH; at.H = H;
    // pruned last application: the attention output of ALL rows goes to a buffer of its own (training: a backward temporary
    // that the attention backward of this application reads again — its slot holds the compact rows), then only the masked
    // rows continue
    const bool pruned_layer = pr != nullptr && l == L - 1;
    bf16_t* const ctx_att = (pruned_layer && stash) ? e->at<bf16_t>(e->o_dy1) : ctx;
    at.scale = 0.125f; at.ctx = ctx_att; at.ldctx = H;
    at.lse = e->at<float>(e->o_lse) + sl * (int64_t)B * e->NH * S;
    // (pruned: nobody reads the context's image of all rows — the compact rows' image is made with the others, below)
    if (f8 && !pruned_layer) { at.ctx8 = c8; at.ldctx8 = H; at.ctx_scale = f8_scale(e, sC); at.ctx_amax = f8_amax(e, sC); }
    TRY(plb_launch_attn_fwd(&at, s));
    if (calib) TRY(plb_launch_amax(ctx_att, 1, (size_t)T, H, H, f8_amax(e, sC), s));
    if (pruned_layer) {
      if (last_application_fwd_pruned(e, pr, stash, calib, tn8, x, ctx_att, Tp, s)) return 1;
      *xout = e->at<bf16_t>(e->o_hm);
      break;
    }
    // dense + residual, LayerNorm
    memset(&g, 0, sizeof(g));
    g.A = ctx; g.lda = H; g.B = e->wbf(PLB_DENSE_W); g.ldb = H; g.M = (int)Tp; g.N = H; g.K = H; g.Mstore = (int)Tp;
    g.bias = e->par(PLB_DENSE_B); g.res = x; g.ldr = H; g.C = pre1; g.ldc = H;
    F8Op od = {c8, e->at<uint8_t>(e->o_wd8), f8_deq(e, sC), f8_deq(e, f8_w(e, F8W_D)), 0};
    PlbLayerNorm ln;
    if (fuse_f) {  // dense + residual + LayerNorm in one launch
      g.C2 = a; g.ldc2 = H;
      ln_fields(e, &g, e->par(PLB_LN1_W), e->par(PLB_LN1_B), e->at<float>(e->o_mean1) + sl * Tp, e->at<float>(e->o_rstd1) + sl * Tp);
      if (f8) f8_out(e, &g, a8, H, sA, 0);
      TRY(gemm_nt_ln_any(&g, 5, f8 ? &od : nullptr, s));
    } else {
      TRY(gemm_nt_any(&g, 0, f8 ? &od : nullptr, s));
      memset(&ln, 0, sizeof(ln));
      ln.x = pre1; ln.ldx = H; ln.gamma = e->par(PLB_LN1_W); ln.beta = e->par(PLB_LN1_B); ln.eps = e->c.layer_norm_eps;
      ln.y = a; ln.ldy = H; ln.T = T; ln.H = H;
      ln.mean = e->at<float>(e->o_mean1) + sl * Tp; ln.rstd = e->at<float>(e->o_rstd1) + sl * Tp;
      if (f8) { ln.out8 = a8; ln.ld8 = H; ln.q_scale = f8_scale(e, sA); ln.q_amax = f8_amax(e, sA); }
      TRY(plb_launch_ln_fwd(&ln, s));
    }
    if (calib) TRY(plb_launch_amax(a, 1, (size_t)T, H, H, f8_amax(e, sA), s));
    // FFN: u = a W1^T + b1, g = gelu_new(u); pre2 = g W2^T + b2 + a
    memset(&g, 0, sizeof(g));
    g.A = a; g.lda = H; g.B = e->wbf(PLB_FFN_W); g.ldb = H; g.M = (int)Tp; g.N = I; g.K = H; g.Mstore = (int)Tp;
    g.bias = e->par(PLB_FFN_B); g.C = u; g.ldc = I; g.C2 = gl; g.ldc2 = I;
    if (f8) f8_out(e, &g, g8, I, sG, 0);
    F8Op o1 = {a8, e->at<uint8_t>(e->o_w18), f8_deq(e, sA), f8_deq(e, f8_w(e, F8W_1)), 0};
    // calls on tile multiples stash gelu_new'(u) in the "u" slot (gelu_dstash): the forward's sigmoid serves the
    // activation and its derivative, and the backward epilogue multiplies instead of evaluating the derivative. In an
    // fp8 call gelu(u) itself leaves as its e4m3 image ALONE: nothing reads it in bf16 (FFN output GEMM and weight
    // gradient take the image)
    if (dstash) {
      if (tn8) { g.C2 = nullptr; g.ldc2 = 0; }
      TRY(gemm_nt_gelud_any(&g, 0, f8 ? &o1 : nullptr, s));
    } else {
      TRY(gemm_nt_any(&g, 1, f8 ? &o1 : nullptr, s));
    }
    if (calib) TRY(plb_launch_amax(gl, 1, (size_t)T, I, I, f8_amax(e, sG), s));
    memset(&g, 0, sizeof(g));
    g.A = gl; g.lda = I; g.B = e->wbf(PLB_FFNO_W); g.ldb = I; g.M = (int)Tp; g.N = H; g.K = I; g.Mstore = (int)Tp;
    g.bias = e->par(PLB_FFNO_B); g.res = a; g.ldr = H; g.C = pre2; g.ldc = H;
    F8Op o2 = {g8, e->at<uint8_t>(e->o_w28), f8_deq(e, sG), f8_deq(e, f8_w(e, F8W_2)), 0};
    const bool next8 = f8 && l + 1 < L;   // the next application's input image
    if (fuse_f) {  // FFN output + residual + LayerNorm in one launch
      g.C2 = y; g.ldc2 = H;
      ln_fields(e, &g, e->par(PLB_LN2_W), e->par(PLB_LN2_B), e->at<float>(e->o_mean2) + sl * Tp, e->at<float>(e->o_rstd2) + sl * Tp);
      if (next8) f8_out(e, &g, x8n, H, f8_site(e, F8_X, l + 1), 0);
      TRY(gemm_nt_ln_any(&g, 5, f8 ? &o2 : nullptr, s));
    } else {
      TRY(gemm_nt_any(&g, 0, f8 ? &o2 : nullptr, s));
      memset(&ln, 0, sizeof(ln));
      ln.x = pre2; ln.ldx = H; ln.gamma = e->par(PLB_LN2_W); ln.beta = e->par(PLB_LN2_B); ln.eps = e->c.layer_norm_eps;
      ln.y = y; ln.ldy = H; ln.T = T; ln.H = H;
      ln.mean = e->at<float>(e->o_mean2) + sl * Tp; ln.rstd = e->at<float>(e->o_rstd2) + sl * Tp;
      if (next8) {
        const int sN = f8_site(e, F8_X, l + 1);
        ln.out8 = x8n; ln.ld8 = H; ln.q_scale = f8_scale(e, sN); ln.q_amax = f8_amax(e, sN);
      }
      TRY(plb_launch_ln_fwd(&ln, s));
    }
    *xout = y;
  }
  return 0;
}

extern "C" int plb_forward(PlbEngine* e, const int64_t* ids, const int32_t* lengths, int32_t B, int32_t S, float* hidden,
                           float* phoneme_logits, float* token_logits, void* stream) {
  if (check_shape(e, B, S, "plb_forward")) return 1;
  if (!ids) return fail("plb_forward: ids is null");
  if (token_logits && !e->NT) return fail("plb_forward: token_logits requested but num_tokens = 0");
  hipStream_t s = (hipStream_t)stream;
  const int H = e->H, T = B * S;
  const int64_t Tp = rup(T, 128);
  bf16_t* x = nullptr;
  if (run_encoder(e, ids, lengths, B, S, false, &x, s)) return 1;
  if (hidden) TRY(plb_launch_bf16_to_f32(x, H, hidden, H, T, H, s));
  PlbGemmNT g;
  if (phoneme_logits) {
    memset(&g, 0, sizeof(g));
    g.A = x; g.lda = H; g.B = e->wbf(PLB_HEAD_W); g.ldb = H; g.M = (int)Tp; g.N = e->NP; g.K = H; g.Mstore = T;
    g.bias = e->par(PLB_HEAD_B); g.Cf = phoneme_logits; g.ldcf = e->NP;
    TRY(plb_launch_gemm_nt(&g, 0, 1, s));
  }
  if (token_logits) {
    memset(&g, 0, sizeof(g));
    g.A = x; g.lda = H; g.B = e->wbf(PLB_TOK_W); g.ldb = H; g.M = (int)Tp; g.N = e->NT; g.K = H; g.Mstore = T;
    g.bias = e->par(PLB_TOK_B); g.Cf = token_logits; g.ldcf = e->NT;
    TRY(plb_launch_gemm_nt(&g, 0, 1, s));
  }
  if (e->fp8_on) {
    TRY(fp8_update_scales(e, s));
    e->fp8_ready = true;
  }
  TRY(plb_launch_step_status(e->at<unsigned int>(e->o_lnerr), nullptr, e->host_err_dev, nullptr, s));
  return 0;
}

extern "C" int plb_pooler(PlbEngine* e, const float* hidden, int32_t B, int32_t S, float* pooled, void* stream) {
  if (!e || !e->ws) return fail("plb_pooler: engine not bound");
  if (!hidden || !pooled || B < 1 || S < 1) return fail("plb_pooler: bad argument");
  TRY(plb_launch_pooler(hidden, B, S, e->H, e->par(PLB_POOL_W), e->par(PLB_POOL_B), pooled, (hipStream_t)stream));
  return 0;
}

// dW[N,K] = A^T B over Mtot rows -> grads[which] (overwrite)
static int weight_grad(PlbEngine* e, const bf16_t* A, int lda, int Ncols, const bf16_t* Bm, int ldb, int64_t Mtot, int N,
                       int K, float* out, hipStream_t s, bool side_slab = false) {
  PlbGemmTN t;
  memset(&t, 0, sizeof(t));
  t.A = A; t.lda = lda; t.Ncols = Ncols; t.B = Bm; t.ldb = ldb; t.Mtot = (int)Mtot; t.N = N; t.K = K;
  t.splits = tn_splits(Mtot, N, K, &t.rows_per_split);
  const bool direct = t.splits == 1 && N == Ncols;  // every element is written exactly once: no slab, no reduce
  if (!direct && (int64_t)t.splits * N * K > (side_slab ? e->slab2_floats : e->slab_floats)) return fail("weight_grad: slab too small");
  t.slab = direct ? out : e->at<float>(side_slab ? e->o_slab2 : e->o_slab);
  if (N == Ncols && tn_big(Mtot, Ncols, K)) {
    const int tok = plb_prof_begin(PLB_K_GEMM_TN, s, 2.0 * (double)Mtot * N * K, 0.0);
    TRY(plb_launch_gemm_tn_big(&t, s));
    plb_prof_end(tok, s);
  } else {
    TRY(plb_launch_gemm_tn(&t, s));
  }
  if (!direct) TRY(plb_launch_reduce_slabs(t.slab, t.splits, (size_t)N * K, out, 0, s));
  return 0;
}

// The same on the per-layer 1-byte images of an fp8 call: A8 = e5m2 gradient image [Mtot, N], B8 = e4m3 activation image
// [Mtot, K] (row strides = widths in bytes), one dequantisation factor per operand site (shared by the L applications).
static int weight_grad8(PlbEngine* e, const uint8_t* A8, const uint8_t* B8, int64_t Mtot, int N, int K, int site_a, int site_b,
                        float* out, hipStream_t s) {
  PlbGemmTN t;
  memset(&t, 0, sizeof(t));
  t.A = reinterpret_cast<const bf16_t*>(A8); t.lda = N; t.Ncols = N; t.B = reinterpret_cast<const bf16_t*>(B8); t.ldb = K;
  t.Mtot = (int)Mtot; t.N = N; t.K = K;
  t.splits = tn_splits(Mtot, N, K, &t.rows_per_split);
  t.rows_per_split = (int)rup(t.rows_per_split, 128);   // K-tiles of 128 tokens
  t.splits = (int)((Mtot + t.rows_per_split - 1) / t.rows_per_split);
  if ((int64_t)t.splits * N * K > e->slab_floats) return fail("weight_grad8: slab too small");
  t.slab = e->at<float>(e->o_slab);
  t.deq_a = f8_deq(e, f8_site(e, site_a, 0)); t.deq_b = f8_deq(e, f8_site(e, site_b, 0));
  const int tok = plb_prof_begin(PLB_K_GEMM_TN_FP8, s, 2.0 * (double)Mtot * N * K, 0.0);
  TRY(plb_launch_gemm_tn_fp8(&t, s));
  plb_prof_end(tok, s);
  TRY(plb_launch_reduce_slabs(t.slab, t.splits, (size_t)N * K, out, 0, s));
  return 0;
}
static bool tn8_ok(const PlbEngine* e, int64_t Mtot) {
  return Mtot % 128 == 0 && e->H % 256 == 0 && e->I % 256 == 0 && Mtot >= 8192;
}

static int backward_tail(PlbEngine* e, const int64_t* masked_ids, bf16_t* dy, int B, int S, int du_rows, hipStream_t s);

// ---- gradient exchange pieces ---------------------------------------------------------------------------------
// One sum all-reduce of grads[a, b) on the communication stream, ordered after everything enqueued on `after` so far.
static int g_debug_skip_piece = -1;   // test hook (plb_debug_skip_piece): drop the n-th piece of a loss call
extern "C" void plb_debug_skip_piece(int index) { g_debug_skip_piece = index; }

static hipEvent_t trace_event(PlbEngine* e) {
  if (!e->trace_pool.empty()) { hipEvent_t v = e->trace_pool.back(); e->trace_pool.pop_back(); return v; }
  hipEvent_t v = nullptr;
  (void)hipEventCreate(&v);   // timing enabled
  return v;
}
static int reduce_piece(PlbEngine* e, int64_t a, int64_t b, hipStream_t after) {
  if (!e->comm || b <= a) return 0;
  if (g_debug_skip_piece >= 0 && e->piece_count == g_debug_skip_piece) {  // what a forgotten tensor looks like
    g_debug_skip_piece = -1;
    e->piece_count += 1;
    return 0;
  }
  PlbEngine::PieceTrace tr{a, b, nullptr, nullptr};
  if (e->trace_on) {
    tr.released = trace_event(e); tr.done = trace_event(e);
    (void)hipEventRecord(tr.released, after);
  }
  HIPTRY(ev_record(e, e->ev_piece, after));
  HIPTRY(ev_wait(e, e->comm_stream, e->ev_piece));
  HB_W(e->comm_stream, e->grads + a, (b - a) * 4, "all-reduce piece (in place)");
  const int rc = g_rccl.AllReduce(e->grads + a, e->grads + a, (size_t)(b - a), kNcclFloat32, kNcclSum, e->comm, e->comm_stream);
  if (rc != kNcclSuccess) return fail("ncclAllReduce: %s", g_rccl.GetErrorString(rc));
  if (e->trace_on) {
    (void)hipEventRecord(tr.done, e->comm_stream);
    e->trace.push_back(tr);
  }
  e->comm_pending = true;
  e->piece_floats += b - a;
  e->piece_count += 1;
  return 0;
}
static bool overlapping(const PlbEngine* e) { return e->comm && e->overlap; }
// Close the pieces issued so far: later joins wait on ev_comm_done.
static int pieces_done(PlbEngine* e) {
  if (e->hb.on && e->hb.violations)
    return fail("happens-before audit: %d violation(s), first: %s", e->hb.violations, e->hb.first.c_str());
  if (!e->comm_pending) return 0;
  // the pieces are disjoint by construction; together they must be exactly the range AdamW is about to consume (a
  // one-rank communicator would not show a forgotten tensor: its all-reduce is the identity)
  const int64_t want = e->ptrain + (e->tok_grads_live ? e->ptotal - e->poff[PLB_TOK_W] : 0);
  if (e->piece_floats != want)
    return fail("gradient exchange covered %lld of %lld floats", (long long)e->piece_floats, (long long)want);
  HIPTRY(ev_record(e, e->ev_comm_done, e->comm_stream));
  e->grads_reduced = true;
  return 0;
}

// ---- the step's health word, agreed between the ranks -----------------------------------------------------------
// A fused LayerNorm hand-off that times out (never observed) raises the error word of THE RANK IT HAPPENED ON; that
// rank's gradients are invalid — and have been summed into every replica by the time AdamW runs. So the word travels
// too: one float per rank (its count), summed over the communicator inside the loss call, after the last launch that
// can raise it (the layer loop; the tail has no hand-offs) and before the call's status launch. Every rank then sees a
// non-zero word, returns a NaN loss, skips the update (and every later one, until plb_status has reported) and raises
// from its next status poll: replicas stay bit-identical. Overlapped form: on the communication stream, between the head
// piece and the first weight's (it is long done when the tail's last GEMM ends; the main stream joins it before the
// status launch); serial form: in the caller's stream. The SAME position in the collective sequence on every rank,
// including a rank that takes the zero-loss path.
static float* status_float(const PlbEngine* e) { return e->at<float>(e->o_lnerr) + 16; }
static int status_exchange(PlbEngine* e, hipStream_t s) {
  if (!e->comm) return 0;
  float* f = status_float(e);
  TRY(plb_launch_status_export(e->at<unsigned int>(e->o_lnerr), f, s));
  hipStream_t cs = overlapping(e) ? e->comm_stream : s;
  if (cs != s) {
    HIPTRY(ev_record(e, e->ev_piece, s));
    HIPTRY(ev_wait(e, cs, e->ev_piece));
  }
  const int rc = g_rccl.AllReduce(f, f, 1, kNcclFloat32, kNcclSum, e->comm, cs);
  if (rc != kNcclSuccess) return fail("ncclAllReduce (status word): %s", g_rccl.GetErrorString(rc));
  if (cs != s) {
    HIPTRY(ev_record(e, e->ev_status, cs));
    e->status_pending = true;
  }
  e->status_collectives += 1;
  return 0;
}
// last launch of a loss call: merge the ranks' word (if it travelled), NaN loss + host mirror when it is set
static int status_finish(PlbEngine* e, float* loss, hipStream_t s) {
  if (e->status_pending) {
    HIPTRY(ev_wait(e, s, e->ev_status));
    e->status_pending = false;
  }
  e->last_loss = loss;
  TRY(plb_launch_step_status(e->at<unsigned int>(e->o_lnerr), loss, e->host_err_dev, e->comm ? status_float(e) : nullptr, s));
  return 0;
}

// token_targets == NULL: the reference's phoneme-only step. Otherwise dual-head: loss = phoneme loss + token loss.
// backward == false: validate() — forward and loss only, one layer of activations, the gradient buffer untouched.
static int loss_impl(PlbEngine* e, bool backward, const int64_t* masked_ids, const int64_t* labels,
                     const int64_t* token_targets, const int32_t* lengths, const int32_t* idx_offsets,
                     const int32_t* idx_flat, int32_t n_masked, int32_t B, int32_t S, float* loss, float* loss_parts,
                     void* stream) {
  const char* who = backward ? "plb_loss_fwd_bwd" : "plb_loss_fwd";
  if (check_shape(e, B, S, who)) return 1;
  if (backward && e->infer) return fail("%s: inference-only engine (PlbConfig.inference_only = 1)", who);
  if (backward && !e->grads) return fail("%s: no gradient buffer bound", who);
  if (!masked_ids || !labels || !idx_offsets || !loss) return fail("%s: null argument", who);
  if (n_masked < 0 || n_masked > e->NMcap) return fail("%s: n_masked %d out of range", who, n_masked);
  if (token_targets && !e->NT) return fail("%s: the engine has no token head (num_tokens = 0)", who);
  hipStream_t s = (hipStream_t)stream;
  const int H = e->H, I = e->I, L = e->L, NP = e->NP;
  const int T = B * S;
  const int64_t Tp = rup(T, 128);
  if (backward) {
    // All-reduce pieces of a PREVIOUS backward that nobody joined (two plb_loss_fwd_bwd calls with no plb_allreduce_grads /
    // plb_adamw_step between them: gradient probing, a caller that skips a step on a bad loss) still read and write the
    // gradient buffer on the communication stream: this call's kernels must not touch it before they have finished.
    if (e->comm && e->comm_pending) HIPTRY(ev_wait(e, s, e->ev_comm_done));
    e->tok_grads_live = token_targets != nullptr;
    e->comm_pending = false;
    e->grads_reduced = false;
    e->piece_floats = 0;
    e->piece_count = 0;
    e->status_collectives = 0;
    if (e->hb.on) {
      // this call's first launches on the caller's stream may touch any byte of the workspace and the gradient buffer:
      // whatever the previous call left running on the side / communication stream must be ordered before them
      HB_W(s, e->ws, e->ws_bytes, "start of a loss call (whole workspace)");
      HB_W(s, e->grads, e->ptotal * 4, "start of a loss call (gradient buffer)");
      if (e->hb.violations) return fail("happens-before audit: %s", e->hb.first.c_str());
      e->hb.new_call();
    }
    for (auto& t : e->trace) { e->trace_pool.push_back(t.released); e->trace_pool.push_back(t.done); }
    e->trace.clear();
    if (e->trace_on) {
      if (!e->tr_call0) { (void)hipEventCreate(&e->tr_call0); (void)hipEventCreate(&e->tr_tail0); (void)hipEventCreate(&e->tr_tail1); }
      (void)hipEventRecord(e->tr_call0, s);
      e->tr_tail_valid = false;
    }
  }
  if (n_masked == 0 && !token_targets) {  // train.py:129 — zero loss, nothing to back-propagate
    HIPTRY(hipMemsetAsync(loss, 0, sizeof(float), s));
    if (backward) {
      HIPTRY(hipMemsetAsync(e->grads, 0, (size_t)e->ptrain * 4, s));
      if (overlapping(e)) {
        // The other ranks still contribute theirs — and they issue the pieces of a regular step: a collective
        // sequence must be the same on every rank, so this rank issues the very same ranges in the very same order
        // (its zeros), not one all-reduce of the whole buffer.
        const int64_t* o = e->poff;
        const int64_t ranges[10][2] = {{o[PLB_HEAD_W], e->ptrain}, {o[PLB_Q_W], o[PLB_Q_B]}, {o[PLB_FFN_W], o[PLB_FFN_B]},
                                       {0, o[PLB_Q_W]}, {o[PLB_Q_B], o[PLB_DENSE_W]}, {o[PLB_DENSE_B], o[PLB_FFN_W]},
                                       {o[PLB_FFN_B], o[PLB_FFNO_W]}, {o[PLB_FFNO_B], o[PLB_HEAD_W]}, {o[PLB_FFNO_W], o[PLB_FFNO_B]},
                                       {o[PLB_DENSE_W], o[PLB_DENSE_B]}};
        for (int i = 0; i < 10; ++i) {
          HB_W(s, e->grads + ranges[i][0], (ranges[i][1] - ranges[i][0]) * 4, "zero gradients of a rank without masked phonemes");
          if (reduce_piece(e, ranges[i][0], ranges[i][1], s)) return 1;
          if (i == 0 && status_exchange(e, s)) return 1;   // where a regular step issues it: behind the head piece
        }
        if (pieces_done(e)) return 1;
      } else if (status_exchange(e, s)) {
        return 1;
      }
      if (status_finish(e, loss, s)) return 1;
    }
    return 0;
  }
  // ---- masked rows: head GEMM, cross-entropy, head gradients ----------------------------------------
  const int NM = (int)rup(n_masked, 128);
  // the row list first: a phoneme-only call runs the post-attention part of its LAST application on these rows alone
  // (last_application_fwd_pruned) when that is less than half of the batch; dual-head calls run every row
  Prune pr = {e->at<int32_t>(e->o_rows), n_masked, NM};
  const bool prune = prune_enabled() && n_masked > 0 && !token_targets && L >= 2 && 2 * (int64_t)NM <= Tp;
  if (n_masked > 0)
    TRY(plb_launch_ce_prepare(idx_offsets, idx_flat, labels, B, S, e->at<int32_t>(e->o_rows), e->at<int32_t>(e->o_tgt),
                              e->at<float>(e->o_w), s));
  if (backward) e->pruned_rows = prune ? NM : 0;
  e->last_call_rows[0] = prune ? NM : Tp; e->last_call_rows[1] = Tp;
  bf16_t* xL = nullptr;
  if (run_encoder(e, masked_ids, lengths, B, S, backward, &xL, s, prune ? &pr : nullptr)) return 1;

  int32_t* rows = e->at<int32_t>(e->o_rows);
  int32_t* tgt = e->at<int32_t>(e->o_tgt);
  float* w = e->at<float>(e->o_w);
  float* lrows = e->at<float>(e->o_lrows);
  bf16_t* hm = e->at<bf16_t>(e->o_hm);
  float* logm = e->at<float>(e->o_logm);
  bf16_t* dlog = e->at<bf16_t>(e->o_dlog);
  bf16_t* dhm = backward ? e->at<bf16_t>(e->o_dhm) : nullptr;
  float* scratch = backward ? e->at<float>(e->o_scratch) : nullptr;
  bf16_t* dy = backward ? e->at<bf16_t>(e->o_dy0) : nullptr;
  bf16_t* dy_other = backward ? e->at<bf16_t>(e->o_dy1) : nullptr;
  PlbGemmNT g;
  if (backward && !prune) HIPTRY(hipMemsetAsync(dy, 0, (size_t)Tp * H * 2, s));
  if (n_masked > 0) {
    if (!prune) TRY(plb_launch_gather_rows(xL, H, rows, n_masked, NM, H, hm, H, s));   // (pruned: xL IS hm, the compact rows)
    memset(&g, 0, sizeof(g));
    g.A = hm; g.lda = H; g.B = e->wbf(PLB_HEAD_W); g.ldb = H; g.M = NM; g.N = NP; g.K = H; g.Mstore = NM;
    g.bias = e->par(PLB_HEAD_B); g.Cf = logm; g.ldcf = 256;
    TRY(plb_launch_gemm_nt(&g, 0, 1, s));
    TRY(plb_launch_ce_fwd_bwd(logm, 256, NP, tgt, w, n_masked, NM, lrows, dlog, 256, s));
    TRY(plb_launch_sum_rows(lrows, n_masked, loss, s));
    if (backward) {
      if (weight_grad(e, dlog, 256, 256, hm, H, NM, NP, H, e->grd(PLB_HEAD_W), s)) return 1;
      TRY(plb_launch_colsum(dlog, 1, (size_t)NM, 256, 256, e->grd(PLB_HEAD_B), NP, 0, scratch, 8, s));
      memset(&g, 0, sizeof(g));
      g.A = dlog; g.lda = 256; g.B = e->at<bf16_t>(e->o_wpT); g.ldb = 256; g.M = NM; g.N = H; g.K = 256; g.Mstore = NM;
      g.C = dhm; g.ldc = H;
      TRY(plb_launch_gemm_nt(&g, 0, 0, s));
      // (pruned: the compact gradient rows dhm ARE the output gradient of the last application's compact part)
      if (!prune) TRY(plb_launch_scatter_rows(dhm, H, rows, n_masked, H, dy, H, s));
    }
  } else {  // dual-head step on a batch without masked phonemes: phoneme loss 0, its head gets zero gradients
    HIPTRY(hipMemsetAsync(loss, 0, sizeof(float), s));
    if (backward)
      HIPTRY(hipMemsetAsync(e->grd(PLB_HEAD_W), 0, (size_t)(e->psize[PLB_HEAD_W] + e->psize[PLB_HEAD_B]) * 4, s));
  }
  // the phoneme head's gradients are final: their all-reduce runs beside the whole backward
  if (backward) HB_W(s, e->grd(PLB_HEAD_W), (e->ptrain - e->poff[PLB_HEAD_W]) * 4, "phoneme head gradients (weight-gradient GEMM, bias column sums)");
  if (backward && overlapping(e) && reduce_piece(e, e->poff[PLB_HEAD_W], e->ptrain, s)) return 1;
  if (loss_parts) HIPTRY(hipMemcpyAsync(loss_parts, loss, sizeof(float), hipMemcpyDeviceToDevice, s));

  // ---- token (grapheme) head over every valid position: fused GEMM + cross-entropy, head gradients, dH ------------
  // The fp32 logits are never stored. Pass 1 computes them tile by tile and keeps, per row and 256-column tile, the
  // maximum and the sum of exponentials (+ the target logit); a small kernel merges those into the row's
  // log-sum-exp, weight and loss; pass 2 recomputes the logits and writes the gradient (softmax - onehot) * w in
  // bf16 [Tp][NTp] (2.1 GB at 16384 x 64000), the operand of dWt = dlogits^T · H and dH = dlogits · Wt, and the
  // column-sum partials that give the bias gradient.
  if (token_targets) {
    const int NT = e->NT, NTp = e->NTp;
    const int tile = (Tp % 256 == 0) ? 256 : 1256;          // 256x256 or 128x256: both 256 columns wide
    const int ntile = NTp / 256, cprows = tile == 256 ? 2 * (int)(Tp / 256) : 2 * (int)(Tp / 128);
    float* tlrows = e->at<float>(e->o_tlrows);
    float* tloss = e->at<float>(e->o_tloss);
    int64_t* ttgt = e->at<int64_t>(e->o_ttgt);
    HIPTRY(hipMemcpyAsync(ttgt, token_targets, (size_t)T * 8, hipMemcpyDeviceToDevice, s));
    if (Tp > T) HIPTRY(hipMemsetAsync(ttgt + T, 0, (size_t)(Tp - T) * 8, s));
    memset(&g, 0, sizeof(g));
    g.A = xL; g.lda = H; g.B = e->wbf(PLB_TOK_W); g.ldb = H; g.M = (int)Tp; g.N = NTp; g.K = H; g.Mstore = (int)Tp;
    g.bias = e->at<float>(e->o_bt);
    g.ce_cols = NT; g.ce_tgt = ttgt;
    g.ce_pmax = e->at<float>(e->o_tpmax); g.ce_psum = e->at<float>(e->o_tpsum); g.ce_tlogit = e->at<float>(e->o_ttl);
    const double ce_flops = 2.0 * (double)Tp * NTp * H;
    int tok = plb_prof_begin(PLB_K_GEMM_NT_CE, s, ce_flops, 0.0);
    TRY(plb_launch_gemm_nt_big(&g, tile, 3, 0, s));
    plb_prof_end(tok, s);
    TRY(plb_launch_token_ce_combine(g.ce_pmax, g.ce_psum, ntile, g.ce_tlogit, lengths, B, S, (int)Tp,
                                    e->at<float>(e->o_tlse), e->at<float>(e->o_tw), tlrows, s));
    TRY(plb_launch_sum_rows(tlrows, T, tloss, s));
    TRY(plb_launch_add_scalar(loss, loss, tloss, s));
    if (loss_parts) HIPTRY(hipMemcpyAsync(loss_parts + 1, tloss, sizeof(float), hipMemcpyDeviceToDevice, s));
    if (backward) {
      bf16_t* tdl = e->at<bf16_t>(e->o_tdl);
      g.ce_lse = e->at<float>(e->o_tlse); g.ce_w = e->at<float>(e->o_tw);
      g.C = tdl; g.ldc = NTp; g.colpart = e->at<float>(e->o_tcolp);
      tok = plb_prof_begin(PLB_K_GEMM_NT_CE, s, ce_flops, 0.0);
      TRY(plb_launch_gemm_nt_big(&g, tile, 4, 0, s));
      plb_prof_end(tok, s);
      TRY(plb_launch_colsum(g.colpart, 0, (size_t)cprows, NTp, NTp, e->grd(PLB_TOK_B), NT, 0, e->at<float>(e->o_tscr), 1, s));
      float* gw = NTp == NT ? e->grd(PLB_TOK_W) : e->at<float>(e->o_tgrad);
      if (weight_grad(e, tdl, NTp, NTp, xL, H, Tp, NTp, H, gw, s)) return 1;
      if (NTp != NT) HIPTRY(hipMemcpyAsync(e->grd(PLB_TOK_W), gw, (size_t)NT * H * 4, hipMemcpyDeviceToDevice, s));
      HB_W(s, e->grd(PLB_TOK_W), (e->ptotal - e->poff[PLB_TOK_W]) * 4, "token head gradients");
      if (overlapping(e) && reduce_piece(e, e->poff[PLB_TOK_W], e->ptotal, s)) return 1;
      // dH += dlogits · Wt, on top of the scattered phoneme-head rows (in place: a tile reads its residual
      // before its own stores)
      memset(&g, 0, sizeof(g));
      g.A = tdl; g.lda = NTp; g.B = e->at<bf16_t>(e->o_wtT); g.ldb = NTp; g.M = (int)Tp; g.N = H; g.K = NTp; g.Mstore = (int)Tp;
      g.res = dy; g.ldr = H; g.C = dy; g.ldc = H;
      TRY(plb_launch_gemm_nt(&g, 0, 0, s));
    }
  }
  if (!backward) {
    if (e->fp8_on) {  // forward-only call in fp8 mode: activation sites only (gradient sites saw nothing and keep theirs)
      TRY(fp8_update_scales(e, s));
      e->fp8_ready = true;
    }
    // (no collective in a loss-only call: ranks may validate different numbers of batches. A word raised here is sticky
    // and travels with the next training call's exchange.)
    e->last_loss = loss;
    TRY(plb_launch_step_status(e->at<unsigned int>(e->o_lnerr), loss, e->host_err_dev, nullptr, s));
    return 0;
  }

  // ---- layers in reverse --------------------------------------------------------------------------------
  // fp8 mode: every dX GEMM reads e5m2 images of its gradient operand — dU = dpre2·W2 and dA = dU·W1 (+ LayerNorm 1
  // backward), dCtx = dpre1·Wd, dX = dQKV·Wqkv (+ LayerNorm 2 backward of the layer below) — written by the launch that
  // produces the gradient (fused LayerNorm-backward / gelu-backward epilogues, the attention-backward kernels, the one
  // standalone LayerNorm backward), one image per layer for the weight-gradient GEMMs at the end.
  const bool f8 = f8_call(e, Tp, true);
  const bool calib = e->fp8_on && !f8;
  // ffn.bias gradient from the dU GEMM's epilogue: 2 partial rows per row tile of the kernel that runs it
  const int du_rows = e->u_is_derivative ? (f8 ? 2 * (int)(Tp / plb_gemm_nt_fp8_gelud_tile_rows((int)Tp)) : 2 * (int)(Tp / 256))
                                         : (f8 ? 2 * (int)(Tp / 128) : plb_gemm_nt_colpart_rows((int)Tp, I, H));
  bf16_t* da = e->at<bf16_t>(e->o_da);
  bf16_t* dctx = e->at<bf16_t>(e->o_dctx);
  // LayerNorm backward inside the dX GEMM that produces its output gradient (gemm_ln.hip). Rows of partials per layer:
  // 2 per 128-row tile in the fused form (the one standalone launch left — LayerNorm 2 of the last application, whose
  // output gradient comes from the head — then uses as many blocks), else the LayerNorm kernel's block count.
  const bool fuse_b = ln_fusable(e, Tp, 2);
  const int prows = fuse_b ? (int)(2 * Tp / 128) : e->ln_blocks;
  e->part_rows_used = prows;
  for (int l = L - 1; l >= 0; --l) {
    bf16_t* qkv = e->at<bf16_t>(e->o_qkv) + (int64_t)l * Tp * 3 * H;
    bf16_t* ctx = e->at<bf16_t>(e->o_ctx) + (int64_t)l * Tp * H;
    bf16_t* pre1 = e->at<bf16_t>(e->o_pre1) + (int64_t)l * Tp * H;
    bf16_t* u = e->at<bf16_t>(e->o_u) + (int64_t)l * Tp * I;
    bf16_t* pre2 = e->at<bf16_t>(e->o_pre2) + (int64_t)l * Tp * H;
    bf16_t* dqkv = e->at<bf16_t>(e->o_dqkv) + (int64_t)l * Tp * 3 * H;
    bf16_t* dpre1 = e->at<bf16_t>(e->o_dpre1) + (int64_t)l * Tp * H;
    bf16_t* du = e->at<bf16_t>(e->o_du) + (int64_t)l * Tp * I;
    bf16_t* dpre2 = e->at<bf16_t>(e->o_dpre2) + (int64_t)l * Tp * H;
    uint8_t* dp8 = e->at<uint8_t>(e->o_dp8) + (int64_t)l * Tp * H;
    uint8_t* du8 = e->at<uint8_t>(e->o_du8) + (int64_t)l * Tp * I;
    uint8_t* dp18 = e->at<uint8_t>(e->o_dp18) + (int64_t)l * Tp * H;
    uint8_t* dq8 = e->at<uint8_t>(e->o_dq8) + (int64_t)l * Tp * 3 * H;
    const int sDP = f8_site(e, F8_DP, l), sDU = f8_site(e, F8_DU, l), sDP1 = f8_site(e, F8_DP1, l), sDQ = f8_site(e, F8_DQ, l);
    PlbLayerNorm ln;
    if (prune && l == L - 1) {
      // ---- backward of the pruned last application: the compact part (LayerNorm 2, FFN, LayerNorm 1, dense) on the Mc
      // masked rows — small-shape launches, the forward's compact activations at the start of this application's slots —
      // then its gradients are scattered back to token rows (zeros elsewhere: that is what the full evaluation computes
      // there) for the attention backward and the dX GEMM, which run on all rows
      const int Mc = NM;
      bf16_t* const dac = e->at<bf16_t>(e->o_da);       // dA of the compact rows, then (full) dpre1 scattered to token rows
      bf16_t* const dctxc = e->at<bf16_t>(e->o_dy0);    // dCtx of the compact rows (dy is not used by this application)
      bf16_t* const ctx_att = e->at<bf16_t>(e->o_dy1);  // the forward's attention output of all rows
      memset(&ln, 0, sizeof(ln));
      ln.x = pre2; ln.ldx = H; ln.gamma = e->par(PLB_LN2_W); ln.T = Mc; ln.H = H; ln.Tzero = Mc;
      ln.mean = e->at<float>(e->o_mean2) + (int64_t)l * Tp; ln.rstd = e->at<float>(e->o_rstd2) + (int64_t)l * Tp;
      ln.dy = dhm; ln.lddy = H; ln.dx = dpre2; ln.lddx = H;
      ln.partials = e->at<float>(e->o_part2) + (int64_t)l * prows * 3 * H; ln.nblocks = prows;
      TRY(plb_launch_ln_bwd(&ln, s));
      if (calib) TRY(plb_launch_amax(dpre2, 1, (size_t)n_masked, H, H, f8_amax(e, sDP), s));
      memset(&g, 0, sizeof(g));
      g.A = dpre2; g.lda = H; g.B = e->at<bf16_t>(e->o_w2T); g.ldb = H; g.M = Mc; g.N = I; g.K = H; g.Mstore = Mc;
      g.aux = u; g.ldaux = I; g.C = du; g.ldc = I;
      TRY(plb_launch_gemm_nt(&g, 2, 0, s));   // (the forward of this part kept u itself: act 1)
      if (calib) TRY(plb_launch_amax(du, 1, (size_t)n_masked, I, I, f8_amax(e, sDU), s));
      if (du_rows > 0) {   // this application's block of ffn.bias partial rows: its column sums in row 0, zeros below
        float* blk = e->at<float>(e->o_ducol) + (int64_t)l * du_rows * I;
        TRY(plb_launch_colsum(du, 1, (size_t)Mc, I, I, blk, I, 0, scratch, 16, s));
        if (du_rows > 1) HIPTRY(hipMemsetAsync(blk + I, 0, (size_t)(du_rows - 1) * I * 4, s));
      }
      memset(&g, 0, sizeof(g));
      g.A = du; g.lda = I; g.B = e->at<bf16_t>(e->o_w1T); g.ldb = I; g.M = Mc; g.N = H; g.K = I; g.Mstore = Mc;
      g.res = dpre2; g.ldr = H; g.C = dac; g.ldc = H;
      TRY(plb_launch_gemm_nt(&g, 0, 0, s));
      memset(&ln, 0, sizeof(ln));
      ln.x = pre1; ln.ldx = H; ln.gamma = e->par(PLB_LN1_W); ln.T = Mc; ln.H = H; ln.Tzero = Mc;
      ln.mean = e->at<float>(e->o_mean1) + (int64_t)l * Tp; ln.rstd = e->at<float>(e->o_rstd1) + (int64_t)l * Tp;
      ln.dy = dac; ln.lddy = H; ln.dx = dpre1; ln.lddx = H;
      ln.partials = e->at<float>(e->o_part1) + (int64_t)l * prows * 3 * H; ln.nblocks = prows;
      TRY(plb_launch_ln_bwd(&ln, s));
      if (calib) TRY(plb_launch_amax(dpre1, 1, (size_t)n_masked, H, H, f8_amax(e, sDP1), s));
      if (e->tn8_call) {   // fp8 call: the compact gradient rows as e5m2 images for the stacked weight-gradient GEMMs
        const void* src[3] = {dpre2, du, dpre1}; const int fl[3] = {3, 3, 3};
        const size_t nel[3] = {(size_t)Mc * H, (size_t)Mc * I, (size_t)Mc * H};
        const float* sc[3] = {f8_scale(e, sDP), f8_scale(e, sDU), f8_scale(e, sDP1)};
        uint8_t* dst[3] = {dp8, du8, dp18};
        float* am[3] = {f8_amax(e, sDP), f8_amax(e, sDU), f8_amax(e, sDP1)};
        TRY(plb_launch_quantize_multi(3, src, fl, nel, sc, dst, am, s));
      }
      memset(&g, 0, sizeof(g));
      g.A = dpre1; g.lda = H; g.B = e->at<bf16_t>(e->o_wdT); g.ldb = H; g.M = Mc; g.N = H; g.K = H; g.Mstore = Mc;
      g.C = dctxc; g.ldc = H;
      TRY(plb_launch_gemm_nt(&g, 0, 0, s));
      // back to token rows: dCtx and dpre1 are zero wherever no masked position sits
      HIPTRY(hipMemsetAsync(dctx, 0, (size_t)Tp * H * 2, s));
      TRY(plb_launch_scatter_rows(dctxc, H, rows, n_masked, H, dctx, H, s));
      HIPTRY(hipMemsetAsync(dac, 0, (size_t)Tp * H * 2, s));   // (dA has been consumed by the LayerNorm backward above)
      TRY(plb_launch_scatter_rows(dpre1, H, rows, n_masked, H, dac, H, s));
      PlbAttn at;
      memset(&at, 0, sizeof(at));
      at.qkv = qkv; at.ldqkv = 3 * H; at.lengths = lengths; at.B = B; at.S = S; at.NH = e->NH; at.H = H; at.scale = 0.125f;
      at.ctx = ctx_att; at.ldctx = H; at.lse = e->at<float>(e->o_lse) + (int64_t)l * B * e->NH * S;
      at.dctx = dctx; at.lddctx = H; at.delta = e->at<float>(e->o_delta); at.dqkv = dqkv; at.lddqkv = 3 * H;
      at.colpart = e->at<float>(e->o_qkvcol) + (int64_t)l * (B * ((S + 127) / 128) * 4) * 3 * H; at.colpart_accumulate = 0;
      if (f8) {   // as in the full evaluation: dQKV's e5m2 image for the fp8 dX GEMM (alone, once the weight gradient reads images too)
        at.dqkv8 = dq8; at.lddqkv8 = 3 * H; at.dqkv_scale = f8_scale(e, sDQ); at.dqkv_amax = f8_amax(e, sDQ);
        if (e->tn8_call) at.dqkv = nullptr;
      }
      TRY(plb_launch_attn_bwd(&at, s));
      if (Tp > T) {
        if (at.dqkv) HIPTRY(hipMemsetAsync(dqkv + (int64_t)T * 3 * H, 0, (size_t)(Tp - T) * 3 * H * 2, s));
        if (f8) HIPTRY(hipMemsetAsync(dq8 + (int64_t)T * 3 * H, 0, (size_t)(Tp - T) * 3 * H, s));
      }
      if (calib) TRY(plb_launch_amax(dqkv, 1, (size_t)T, 3 * H, 3 * H, f8_amax(e, sDQ), s));
      // dX = dQKV · Wqkv + dpre1 (token rows), with the LayerNorm-2 backward of application L-2 in the epilogue where fused
      memset(&g, 0, sizeof(g));
      g.A = dqkv; g.lda = 3 * H; g.B = e->at<bf16_t>(e->o_wqkvT); g.ldb = 3 * H; g.M = (int)Tp; g.N = H; g.K = 3 * H;
      g.Mstore = (int)Tp; g.res = dac; g.ldr = H; g.C = dy_other; g.ldc = H;
      F8Op oxp = {dq8, e->at<uint8_t>(e->o_wqT8), f8_deq(e, sDQ), f8_deq(e, f8_w(e, F8W_QKVT)), 1};
      if (fuse_b) {
        g.C = e->at<bf16_t>(e->o_dpre2) + (int64_t)(l - 1) * Tp * H;
        g.aux = e->at<bf16_t>(e->o_pre2) + (int64_t)(l - 1) * Tp * H; g.ldaux = H;
        g.colpart = e->at<float>(e->o_part2) + (int64_t)(l - 1) * prows * 3 * H;
        ln_fields(e, &g, e->par(PLB_LN2_W), nullptr, e->at<float>(e->o_mean2) + (int64_t)(l - 1) * Tp, e->at<float>(e->o_rstd2) + (int64_t)(l - 1) * Tp);
        if (f8) f8_out(e, &g, e->at<uint8_t>(e->o_dp8) + (int64_t)(l - 1) * Tp * H, H, f8_site(e, F8_DP, l - 1), 1);
        TRY(gemm_nt_ln_any(&g, 6, f8 ? &oxp : nullptr, s));
      } else {
        TRY(gemm_nt_any(&g, 0, f8 ? &oxp : nullptr, s));
      }
      bf16_t* tmp = dy; dy = dy_other; dy_other = tmp;
      continue;
    }
    if (!(fuse_b && l != L - 1)) {  // fused form: the dX GEMM of application l+1 wrote dpre2 of this one (see below)
      memset(&ln, 0, sizeof(ln));
      ln.x = pre2; ln.ldx = H; ln.gamma = e->par(PLB_LN2_W); ln.T = T; ln.H = H; ln.Tzero = (int)Tp;
      ln.mean = e->at<float>(e->o_mean2) + (int64_t)l * Tp; ln.rstd = e->at<float>(e->o_rstd2) + (int64_t)l * Tp;
      ln.dy = dy; ln.lddy = H; ln.dx = dpre2; ln.lddx = H;
      ln.partials = e->at<float>(e->o_part2) + (int64_t)l * prows * 3 * H; ln.nblocks = prows;
      if (f8) { ln.out8 = dp8; ln.ld8 = H; ln.q_scale = f8_scale(e, sDP); ln.q_amax = f8_amax(e, sDP); }
      TRY(plb_launch_ln_bwd(&ln, s));
    }
    if (calib) TRY(plb_launch_amax(dpre2, 1, (size_t)T, H, H, f8_amax(e, sDP), s));
    // dU = (dpre2 · W2) ∘ gelu'(u)
    memset(&g, 0, sizeof(g));
    g.A = dpre2; g.lda = H; g.B = e->at<bf16_t>(e->o_w2T); g.ldb = H; g.M = (int)Tp; g.N = I; g.K = H; g.Mstore = (int)Tp;
    g.aux = u; g.ldaux = I; g.C = du; g.ldc = I;
    if (du_rows > 0) g.colpart = e->at<float>(e->o_ducol) + (int64_t)l * du_rows * I;
    if (f8) f8_out(e, &g, du8, I, sDU, 1);
    F8Op ou = {dp8, e->at<uint8_t>(e->o_w2T8), f8_deq(e, sDP), f8_deq(e, f8_w(e, F8W_2T)), 1};
    if (e->u_is_derivative) {   // what the forward of THIS call stashed; fp8: dU leaves as its e5m2 image alone
      if (e->tn8_call) g.C = nullptr;
      TRY(gemm_nt_gelud_any(&g, 1, f8 ? &ou : nullptr, s));
    } else {
      TRY(gemm_nt_any(&g, 2, f8 ? &ou : nullptr, s));
    }
    if (calib) TRY(plb_launch_amax(du, 1, (size_t)T, I, I, f8_amax(e, sDU), s));
    // dA = dU · W1 + dpre2
    memset(&g, 0, sizeof(g));
    g.A = du; g.lda = I; g.B = e->at<bf16_t>(e->o_w1T); g.ldb = I; g.M = (int)Tp; g.N = H; g.K = I; g.Mstore = (int)Tp;
    g.res = dpre2; g.ldr = H; g.C = da; g.ldc = H;
    F8Op oa = {du8, e->at<uint8_t>(e->o_w1T8), f8_deq(e, sDU), f8_deq(e, f8_w(e, F8W_1T)), 1};
    if (fuse_b) {
      // dA = dU · W1 + dpre2 is the gradient of LayerNorm 1's output: its backward runs in this GEMM's epilogue and dA
      // is never stored (dpre1 = the gradient of the LayerNorm's input, + the dgamma | dbeta | bias-gradient partials)
      g.C = dpre1; g.aux = pre1; g.ldaux = H;
      g.colpart = e->at<float>(e->o_part1) + (int64_t)l * prows * 3 * H;
      ln_fields(e, &g, e->par(PLB_LN1_W), nullptr, e->at<float>(e->o_mean1) + (int64_t)l * Tp, e->at<float>(e->o_rstd1) + (int64_t)l * Tp);
      if (f8) f8_out(e, &g, dp18, H, sDP1, 1);
      TRY(gemm_nt_ln_any(&g, 6, f8 ? &oa : nullptr, s));
    } else {
      TRY(gemm_nt_any(&g, 0, f8 ? &oa : nullptr, s));
      memset(&ln, 0, sizeof(ln));
      ln.x = pre1; ln.ldx = H; ln.gamma = e->par(PLB_LN1_W); ln.T = T; ln.H = H; ln.Tzero = (int)Tp;
      ln.mean = e->at<float>(e->o_mean1) + (int64_t)l * Tp; ln.rstd = e->at<float>(e->o_rstd1) + (int64_t)l * Tp;
      ln.dy = da; ln.lddy = H; ln.dx = dpre1; ln.lddx = H;
      ln.partials = e->at<float>(e->o_part1) + (int64_t)l * prows * 3 * H; ln.nblocks = prows;
      if (f8) { ln.out8 = dp18; ln.ld8 = H; ln.q_scale = f8_scale(e, sDP1); ln.q_amax = f8_amax(e, sDP1); }
      TRY(plb_launch_ln_bwd(&ln, s));
    }
    if (calib) TRY(plb_launch_amax(dpre1, 1, (size_t)T, H, H, f8_amax(e, sDP1), s));
    // dCtx = dpre1 · Wd
    memset(&g, 0, sizeof(g));
    g.A = dpre1; g.lda = H; g.B = e->at<bf16_t>(e->o_wdT); g.ldb = H; g.M = (int)Tp; g.N = H; g.K = H; g.Mstore = (int)Tp;
    g.C = dctx; g.ldc = H;
    F8Op oc = {dp18, e->at<uint8_t>(e->o_wdT8), f8_deq(e, sDP1), f8_deq(e, f8_w(e, F8W_DT)), 1};
    TRY(gemm_nt_any(&g, 0, f8 ? &oc : nullptr, s));
    PlbAttn at;
    memset(&at, 0, sizeof(at));
    at.qkv = qkv; at.ldqkv = 3 * H; at.lengths = lengths; at.B = B; at.S = S; at.NH = e->NH; at.H = H; at.scale = 0.125f;
    at.ctx = ctx; at.ldctx = H; at.lse = e->at<float>(e->o_lse) + (int64_t)l * B * e->NH * S;
    at.dctx = dctx; at.lddctx = H; at.delta = e->at<float>(e->o_delta); at.dqkv = dqkv; at.lddqkv = 3 * H;
    at.colpart = e->at<float>(e->o_qkvcol) + (int64_t)l * (B * ((S + 127) / 128) * 4) * 3 * H; at.colpart_accumulate = 0;
    if (f8) {   // dQKV leaves as its e5m2 image (alone, once the weight gradient reads images too)
      at.dqkv8 = dq8; at.lddqkv8 = 3 * H; at.dqkv_scale = f8_scale(e, sDQ); at.dqkv_amax = f8_amax(e, sDQ);
      if (e->tn8_call) at.dqkv = nullptr;
    }
    TRY(plb_launch_attn_bwd(&at, s));
    if (Tp > T) {
      if (at.dqkv) HIPTRY(hipMemsetAsync(dqkv + (int64_t)T * 3 * H, 0, (size_t)(Tp - T) * 3 * H * 2, s));
      if (f8) HIPTRY(hipMemsetAsync(dq8 + (int64_t)T * 3 * H, 0, (size_t)(Tp - T) * 3 * H, s));
    }
    if (calib) TRY(plb_launch_amax(dqkv, 1, (size_t)T, 3 * H, 3 * H, f8_amax(e, sDQ), s));
    // dX = dQKV · Wqkv + dpre1
    memset(&g, 0, sizeof(g));
    g.A = dqkv; g.lda = 3 * H; g.B = e->at<bf16_t>(e->o_wqkvT); g.ldb = 3 * H; g.M = (int)Tp; g.N = H; g.K = 3 * H;
    g.Mstore = (int)Tp; g.res = dpre1; g.ldr = H; g.C = dy_other; g.ldc = H;
    F8Op ox = {dq8, e->at<uint8_t>(e->o_wqT8), f8_deq(e, sDQ), f8_deq(e, f8_w(e, F8W_QKVT)), 1};
    if (fuse_b && l > 0) {
      // the gradient of this application's input is the gradient of LayerNorm 2's output of application l-1: that
      // LayerNorm's backward runs here and writes dpre2 of application l-1 directly
      g.C = e->at<bf16_t>(e->o_dpre2) + (int64_t)(l - 1) * Tp * H;
      g.aux = e->at<bf16_t>(e->o_pre2) + (int64_t)(l - 1) * Tp * H; g.ldaux = H;
      g.colpart = e->at<float>(e->o_part2) + (int64_t)(l - 1) * prows * 3 * H;
      ln_fields(e, &g, e->par(PLB_LN2_W), nullptr, e->at<float>(e->o_mean2) + (int64_t)(l - 1) * Tp, e->at<float>(e->o_rstd2) + (int64_t)(l - 1) * Tp);
      if (f8) f8_out(e, &g, e->at<uint8_t>(e->o_dp8) + (int64_t)(l - 1) * Tp * H, H, f8_site(e, F8_DP, l - 1), 1);
      TRY(gemm_nt_ln_any(&g, 6, f8 ? &ox : nullptr, s));
    } else {
      TRY(gemm_nt_any(&g, 0, f8 ? &ox : nullptr, s));
    }
    bf16_t* tmp = dy; dy = dy_other; dy_other = tmp;
  }
  // the last launch that can raise the hand-off error word is behind us: the word travels now (beside the tail)
  if (status_exchange(e, s)) return 1;
  if (backward_tail(e, masked_ids, dy, B, S, du_rows, s)) return 1;
  if (e->fp8_on) {
    // This call's maxima become the next call's scales; a calibration call arms the fp8 path. AFTER the tail: the weight-
    // gradient GEMMs dequantise this call's images with the scales they were written with (updated before the tail, a
    // call that follows one with 4x larger gradients came out 2x off: tools/fp8_diag.py).
    TRY(fp8_update_scales(e, s));
    e->fp8_ready = true;
    e->fp8_bwd_ready = true;
  }
  // Last launch of the step: a hand-off of the fused LayerNorm launches that timed out — on ANY rank — turns the loss into
  // NaN and shows in plb_poll_status; plb_adamw_step skips on the same word. No host round trip anywhere.
  return status_finish(e, loss, s);
}

// Tail of the backward on two streams.
//  main: the four large token-major weight-gradient GEMMs (MFMA-bound, ~2 ms at config A), each followed — when a
//        communicator is attached — by the all-reduce of the weight it completed, on the communication stream: weight i
//        travels over xGMI while GEMM i+1 runs. The smallest GEMM goes last, so only dense.weight and the small
//        tensors (3.2 MB of 23.4) have nothing left to hide behind.
//  side: everything else that only needs finished gradients — embedding chain, bias and LayerNorm-affine column sums
//        (HBM-bound) — with its own slab / scratch so nothing is shared; joined before the first piece that holds
//        any of its outputs.
static int backward_tail_streams(PlbEngine* e, const int64_t* masked_ids, bf16_t* dy, int B, int S, int du_rows,
                                 hipStream_t s, hipStream_t s2, float* scratch2) {
  const int E = e->E, H = e->H, I = e->I, L = e->L;
  const int T = B * S;
  const int64_t Tp = rup(T, 128);
  const int64_t Mtot = (int64_t)L * Tp;
  // stacked rows of the operands whose last application ran on its masked rows only (ffn.weight, ffn_output.weight,
  // dense.weight: their slots of application L-1 hold Mc compact rows); the Q/K/V weights' operands are always full
  const int64_t Mtot_c = e->pruned_rows ? (int64_t)(L - 1) * Tp + e->pruned_rows : Mtot;
  PlbGemmNT g;
  // side stream -------------------------------------------------------------------------------------------------------
  // (HB_R / HB_W: the happens-before audit's view of each launch — what it reads that another stream wrote, what it
  // writes that another stream reads. The stash operands of the GEMMs are only ever written in the layer loop, which the
  // fork orders before both streams: they are covered by the whole-workspace entry at the fork.)
  bf16_t* evec = e->at<bf16_t>(e->o_e);
  bf16_t* de = e->at<bf16_t>(e->o_de);
  memset(&g, 0, sizeof(g));
  g.A = dy; g.lda = H; g.B = e->at<bf16_t>(e->o_winT); g.ldb = H; g.M = (int)Tp; g.N = E; g.K = H; g.Mstore = (int)Tp;
  g.C = de; g.ldc = E;
  HB_R(s2, dy, Tp * H * 2, "dX of application 0"); HB_W(s2, de, Tp * E * 2, "dE (map-in backward)");
  TRY(plb_launch_gemm_nt(&g, 0, 0, s2));
  HB_W(s2, e->at<float>(s2 != s ? e->o_slab2 : e->o_slab), (s2 != s ? e->slab2_floats : e->slab_floats) * 4, "map-in weight-gradient slab");
  HB_W(s2, e->grd(PLB_MAP_W), e->psize[PLB_MAP_W] * 4, "map-in weight gradient");
  if (weight_grad(e, dy, H, H, evec, E, Tp, H, E, e->grd(PLB_MAP_W), s2, s2 != s)) return 1;
  HB_W(s2, scratch2, 512 * (3 * H > I ? 3 * H : I) * 4, "column-sum scratch of the side stream");
  HB_W(s2, e->grd(PLB_MAP_B), H * 4, "map-in bias gradient");
  TRY(plb_launch_colsum(dy, 1, (size_t)Tp, H, H, e->grd(PLB_MAP_B), H, 0, scratch2, 128, s2));
  HB_W(s2, e->grd(PLB_TYPE_EMB), e->psize[PLB_TYPE_EMB] * 4, "token-type embedding gradient");
  HIPTRY(hipMemsetAsync(e->grd(PLB_TYPE_EMB), 0, (size_t)e->psize[PLB_TYPE_EMB] * 4, s2));
  PlbEmbed em;
  memset(&em, 0, sizeof(em));
  em.ids = masked_ids; em.T = T; em.S = S; em.E = E; em.V = e->V;
  em.word = e->par(PLB_WORD_EMB); em.pos = e->par(PLB_POS_EMB); em.type0 = e->par(PLB_TYPE_EMB);
  em.gamma = e->par(PLB_EMB_LN_W); em.beta = e->par(PLB_EMB_LN_B); em.eps = e->c.layer_norm_eps;
  em.dout = de; em.lddo = E; em.dword = e->grd(PLB_WORD_EMB); em.dpos = e->grd(PLB_POS_EMB);
  em.dx = e->at<float>(e->o_dxe);
  em.partials = e->at<float>(e->o_parte); em.nblocks = e->emb_blocks;
  HB_W(s2, em.dx, Tp * E * 4, "embedding LayerNorm backward rows"); HB_W(s2, em.partials, (int64_t)e->emb_blocks * 2 * E * 4, "embedding LayerNorm partials");
  HB_W(s2, e->grd(PLB_WORD_EMB), (e->poff[PLB_MAP_W] - e->poff[PLB_WORD_EMB]) * 4, "embedding tables' and embedding LayerNorm's gradients");
  TRY(plb_launch_embed_bwd(&em, s2));
  TRY(plb_launch_embed_scatter(&em, e->P, s2));
  TRY(plb_launch_colsum(em.partials, 0, (size_t)e->emb_blocks, 2 * E, 2 * E, e->grd(PLB_EMB_LN_W), 2 * E, 0, scratch2, 1, s2));
  // token_type row 0 receives every token's gradient = the column sums of dpos
  TRY(plb_launch_colsum(e->grd(PLB_POS_EMB), 0, (size_t)e->P, E, E, e->grd(PLB_TYPE_EMB), E, 0, scratch2, 1, s2));
  // Q/K/V biases: the attention-backward kernels left the column sums of every 32-row patch they stored, per application
  // ([L][B*QT*4][3H])
  HB_R(s2, e->at<float>(e->o_qkvcol), (int64_t)L * (B * ((S + 127) / 128) * 4) * 3 * H * 4, "Q/K/V bias partial rows");
  HB_W(s2, e->grd(PLB_Q_B), 3 * H * 4, "Q/K/V bias gradients");
  TRY(plb_launch_colsum(e->at<float>(e->o_qkvcol), 0, (size_t)L * (size_t)(B * ((S + 127) / 128) * 4), 3 * H, 3 * H, e->grd(PLB_Q_B),
                        3 * H, 0, scratch2, 64, s2));
  HB_W(s2, e->grd(PLB_FFN_B), I * 4, "ffn.bias gradient");
  if (du_rows > 0) {
    HB_R(s2, e->at<float>(e->o_ducol), (int64_t)L * du_rows * I * 4, "dU column-sum partial rows");
    TRY(plb_launch_colsum(e->at<float>(e->o_ducol), 0, (size_t)L * du_rows, I, I, e->grd(PLB_FFN_B), I, 0, scratch2, 16, s2));
  } else {
    HB_R(s2, e->at<bf16_t>(e->o_du), Mtot_c * I * 2, "dU of every application");
    TRY(plb_launch_colsum(e->at<bf16_t>(e->o_du), 1, (size_t)Mtot_c, I, I, e->grd(PLB_FFN_B), I, 0, scratch2, 64, s2));
  }
  // LayerNorm-backward partials [L*blocks][3H]: dgamma | dbeta | column sums of dx. (Summing the L applications into
  // one image inside the kernel — PlbLayerNorm.accumulate — was measured: the read-modify-write costs the main stream
  // 2.5 us per launch to save side-stream traffic that is hidden behind the weight-gradient GEMMs anyway.) The third block is the bias
  // gradient of the Linear that produced the LayerNorm's input (dense.bias = colsum(dpre1), ffn_output.bias =
  // colsum(dpre2)): no pass over the stacked gradients.
  const size_t prow = (size_t)L * e->part_rows_used;
  HB_R(s2, e->at<float>(e->o_part1), (int64_t)L * e->part_rows * 3 * H * 4, "LayerNorm-1 backward partial rows");
  HB_W(s2, e->grd(PLB_DENSE_B), 3 * H * 4, "dense.bias + LayerNorm-1 affine gradients");
  TRY(plb_launch_colsum(e->at<float>(e->o_part1), 0, prow, 3 * H, 3 * H, e->grd(PLB_LN1_W), 2 * H, 0, scratch2, 64, s2));
  TRY(plb_launch_copy_cols(scratch2, 64, 3 * H, 2 * H, H, e->grd(PLB_DENSE_B), s2));
  HB_R(s2, e->at<float>(e->o_part2), (int64_t)L * e->part_rows * 3 * H * 4, "LayerNorm-2 backward partial rows");
  HB_W(s2, e->grd(PLB_LN2_W), 2 * H * 4, "LayerNorm-2 affine gradients"); HB_W(s2, e->grd(PLB_FFNO_B), H * 4, "ffn_output.bias gradient");
  TRY(plb_launch_colsum(e->at<float>(e->o_part2), 0, prow, 3 * H, 3 * H, e->grd(PLB_LN2_W), 2 * H, 0, scratch2, 64, s2));
  TRY(plb_launch_copy_cols(scratch2, 64, 3 * H, 2 * H, H, e->grd(PLB_FFNO_B), s2));
  if (s2 != s) HIPTRY(ev_record(e, e->ev_join, s2));
  // main stream: shared-layer weight gradients, one token-major GEMM per weight over all L applications ------------
  // Overlapped exchange: a weight's range travels as soon as its GEMM (+ slab reduction) has written it; the small
  // tensors between the weights in the flat order (biases, LayerNorm, embeddings) travel behind the SIDE stream's event
  // (below, after the first weight's piece); the smallest weight goes last.
  const bool ov = overlapping(e);
  const bool t8 = e->tn8_call;   // fp8 call: gradient (e5m2) x activation (e4m3) images of all L applications
  float* const slab = e->at<float>(e->o_slab);
  HB_W(s, slab, e->slab_floats * 4, "weight-gradient slab"); HB_W(s, e->grd(PLB_Q_W), 3 * H * H * 4, "Q/K/V weight gradients");
  if (t8 ? weight_grad8(e, e->at<uint8_t>(e->o_dq8), e->at<uint8_t>(e->o_x8), Mtot, 3 * H, H, F8_DQ, F8_X, e->grd(PLB_Q_W), s)
         : weight_grad(e, e->at<bf16_t>(e->o_dqkv), 3 * H, 3 * H, e->at<bf16_t>(e->o_x), H, Mtot, 3 * H, H, e->grd(PLB_Q_W), s)) return 1;
  if (ov && reduce_piece(e, e->poff[PLB_Q_W], e->poff[PLB_Q_B], s)) return 1;
  HB_W(s, slab, e->slab_floats * 4, "weight-gradient slab"); HB_W(s, e->grd(PLB_FFN_W), (int64_t)I * H * 4, "ffn.weight gradient");
  if (t8 ? weight_grad8(e, e->at<uint8_t>(e->o_du8), e->at<uint8_t>(e->o_a8), Mtot_c, I, H, F8_DU, F8_A, e->grd(PLB_FFN_W), s)
         : weight_grad(e, e->at<bf16_t>(e->o_du), I, I, e->at<bf16_t>(e->o_a), H, Mtot_c, I, H, e->grd(PLB_FFN_W), s)) return 1;
  if (ov && reduce_piece(e, e->poff[PLB_FFN_W], e->poff[PLB_FFN_B], s)) return 1;
  if (ov) {
    // The small tensors between the weights in the flat order (embeddings + map-in + LayerNorm 2 | Q/K/V biases | dense.bias +
    // LayerNorm 1 | ffn.bias | ffn_output.bias) all come from the side stream, which is done after about three of the four
    // GEMMs: their pieces are released by the SIDE stream's own event (everything it does in this call has been enqueued
    // above), behind the second weight's piece in the communication stream's queue (the side stream, stretched by the
    // GEMMs it runs beside, ends between GEMM 2 and GEMM 3: piece_trace) — five latency-bound all-reduces that
    // travel beside the remaining GEMMs instead of after the last one (they were the step's exposed tail at world > 1:
    // four collectives in a row behind the join). The main stream joins the side stream at the end of the tail as before.
    const int64_t* o = e->poff;
    if (reduce_piece(e, 0, o[PLB_Q_W], s2)) return 1;
    if (reduce_piece(e, o[PLB_Q_B], o[PLB_DENSE_W], s2)) return 1;
    if (reduce_piece(e, o[PLB_DENSE_B], o[PLB_FFN_W], s2)) return 1;
    if (reduce_piece(e, o[PLB_FFN_B], o[PLB_FFNO_W], s2)) return 1;
    if (reduce_piece(e, o[PLB_FFNO_B], o[PLB_HEAD_W], s2)) return 1;
  }
  HB_W(s, slab, e->slab_floats * 4, "weight-gradient slab"); HB_W(s, e->grd(PLB_FFNO_W), (int64_t)I * H * 4, "ffn_output.weight gradient");
  if (t8 ? weight_grad8(e, e->at<uint8_t>(e->o_dp8), e->at<uint8_t>(e->o_g8), Mtot_c, H, I, F8_DP, F8_G, e->grd(PLB_FFNO_W), s)
         : weight_grad(e, e->at<bf16_t>(e->o_dpre2), H, H, e->at<bf16_t>(e->o_g), I, Mtot_c, H, I, e->grd(PLB_FFNO_W), s)) return 1;
  if (ov && reduce_piece(e, e->poff[PLB_FFNO_W], e->poff[PLB_FFNO_B], s)) return 1;
  HB_W(s, slab, e->slab_floats * 4, "weight-gradient slab"); HB_W(s, e->grd(PLB_DENSE_W), (int64_t)H * H * 4, "dense.weight gradient");
  if (t8 ? weight_grad8(e, e->at<uint8_t>(e->o_dp18), e->at<uint8_t>(e->o_c8), Mtot_c, H, H, F8_DP1, F8_C, e->grd(PLB_DENSE_W), s)
         : weight_grad(e, e->at<bf16_t>(e->o_dpre1), H, H, e->at<bf16_t>(e->o_ctx), H, Mtot_c, H, H, e->grd(PLB_DENSE_W), s)) return 1;
  if (ov && reduce_piece(e, e->poff[PLB_DENSE_W], e->poff[PLB_DENSE_B], s)) return 1;   // the smallest weight goes last
  return 0;
}

static int backward_tail(PlbEngine* e, const int64_t* masked_ids, bf16_t* dy, int B, int S, int du_rows, hipStream_t s) {
  hipStream_t s2 = s;
  float* scratch2 = e->at<float>(e->o_scratch);
  // the layer loop (all of it on the caller's stream) has written the stash, the partial-row tables, dX: one entry
  HB_W(s, e->ws, e->ws_bytes, "layer loop (whole workspace)");
  if (e->trace_on) (void)hipEventRecord(e->tr_tail0, s);
  if (e->side) {
    s2 = e->side;
    scratch2 = e->at<float>(e->o_scratch2);
    HIPTRY(ev_record(e, e->ev_fork, s));
    HIPTRY(ev_wait(e, s2, e->ev_fork));
  }
  const int rc = backward_tail_streams(e, masked_ids, dy, B, S, du_rows, s, s2, scratch2);
  // Whatever happened above, the caller's stream must not run ahead of the side stream's work (also on an error
  // path: the side stream may hold launches that read buffers the caller is about to reuse).
  if (s2 != s) {
    if (rc) (void)ev_record(e, e->ev_join, s2);
    const hipError_t je = ev_wait(e, s, e->ev_join);
    if (!rc && je != hipSuccess) return fail("plb_loss_fwd_bwd: joining the side stream: %s", hipGetErrorString(je));
  }
  if (e->trace_on) { (void)hipEventRecord(e->tr_tail1, s); e->tr_tail_valid = true; }
  if (rc) return rc;
  // from here on the caller's stream may again touch anything in the workspace (the next call's forward will)
  HB_W(s, e->ws, e->ws_bytes, "after the side stream's join (whole workspace)");
  return pieces_done(e);
}

extern "C" int plb_loss_fwd_bwd(PlbEngine* e, const int64_t* masked_ids, const int64_t* labels, const int32_t* lengths,
                                const int32_t* idx_offsets, const int32_t* idx_flat, int32_t n_masked, int32_t B,
                                int32_t S, float* loss, void* stream) {
  return loss_impl(e, true, masked_ids, labels, nullptr, lengths, idx_offsets, idx_flat, n_masked, B, S, loss, nullptr, stream);
}

extern "C" int plb_loss_fwd_bwd_dual(PlbEngine* e, const int64_t* masked_ids, const int64_t* labels,
                                     const int64_t* token_ids, const int32_t* lengths, const int32_t* idx_offsets,
                                     const int32_t* idx_flat, int32_t n_masked, int32_t B, int32_t S, float* loss,
                                     float* loss_parts, void* stream) {
  if (!token_ids) return fail("plb_loss_fwd_bwd_dual: token_ids is null");
  return loss_impl(e, true, masked_ids, labels, token_ids, lengths, idx_offsets, idx_flat, n_masked, B, S, loss, loss_parts,
                   stream);
}

extern "C" int plb_loss_fwd(PlbEngine* e, const int64_t* masked_ids, const int64_t* labels, const int64_t* token_ids,
                            const int32_t* lengths, const int32_t* idx_offsets, const int32_t* idx_flat, int32_t n_masked,
                            int32_t B, int32_t S, float* loss, float* loss_parts, void* stream) {
  return loss_impl(e, false, masked_ids, labels, token_ids, lengths, idx_offsets, idx_flat, n_masked, B, S, loss, loss_parts,
                   stream);
}

// ---- data-parallel exchange -------------------------------------------------------------------------------------
extern "C" int plb_comm_unique_id(uint8_t id[PLB_COMM_ID_BYTES]) {
  if (!id) return fail("plb_comm_unique_id: null argument");
  if (const char* err = rccl_load()) return fail("plb_comm_unique_id: %s", err);
  RcclId u;
  memset(&u, 0, sizeof(u));
  const int rc = g_rccl.GetUniqueId(&u);
  if (rc != kNcclSuccess) return fail("ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
  static_assert(sizeof(u) == PLB_COMM_ID_BYTES, "unique id size");
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int plb_comm_destroy(PlbEngine* e) {
  if (!e) return fail("plb_comm_destroy: null engine");
  if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
  if (e->comm && g_rccl.ok) (void)g_rccl.CommDestroy(e->comm);
  e->comm = nullptr; e->comm_rank = 0; e->comm_world = 1; e->comm_pending = false;
  if (e->ev_piece) { (void)hipEventDestroy(e->ev_piece); e->ev_piece = nullptr; }
  if (e->ev_comm_done) { (void)hipEventDestroy(e->ev_comm_done); e->ev_comm_done = nullptr; }
  if (e->ev_status) { (void)hipEventDestroy(e->ev_status); e->ev_status = nullptr; }
  e->status_pending = false;
  if (e->comm_stream) { (void)hipStreamDestroy(e->comm_stream); e->comm_stream = nullptr; }
  return 0;
}

extern "C" int plb_comm_init(PlbEngine* e, const uint8_t id[PLB_COMM_ID_BYTES], int32_t rank, int32_t world) {
  if (!e || !id) return fail("plb_comm_init: null argument");
  if (!e->grads) return fail("plb_comm_init: bind the gradient buffer first (plb_bind)");
  if (world < 1 || rank < 0 || rank >= world) return fail("plb_comm_init: rank %d of %d", rank, world);
  if (e->comm) return fail("plb_comm_init: the engine already has a communicator");
  if (const char* err = rccl_load()) return fail("plb_comm_init: %s", err);
  // priority stream: the collective's few workgroups should get CUs ahead of the next GEMM's grid
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  HIPTRY(hipStreamCreateWithPriority(&e->comm_stream, hipStreamNonBlocking, hi));
  HIPTRY(hipEventCreateWithFlags(&e->ev_piece, kStreamOrderEvent));
  HIPTRY(hipEventCreateWithFlags(&e->ev_comm_done, kStreamOrderEvent));
  HIPTRY(hipEventCreateWithFlags(&e->ev_status, kStreamOrderEvent));
  RcclId u;
  memcpy(&u, id, sizeof(u));
  const int rc = g_rccl.CommInitRank(&e->comm, world, u, rank);
  if (rc != kNcclSuccess) {
    e->comm = nullptr;
    (void)plb_comm_destroy(e);
    return fail("ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(rc));
  }
  e->comm_rank = rank; e->comm_world = world;
  return 0;
}

extern "C" int plb_comm_info(const PlbEngine* e, int32_t* rank, int32_t* world, int32_t* rccl_version) {
  if (!e) return fail("plb_comm_info: null engine");
  if (rank) *rank = e->comm_rank;
  if (world) *world = e->comm ? e->comm_world : 1;
  if (rccl_version) {
    int v = 0;
    if (g_rccl.ok) (void)g_rccl.GetVersion(&v);
    *rccl_version = v;
  }
  return 0;
}

extern "C" int plb_status_ex(PlbEngine* e, int32_t* ln_exchange_timeouts, int32_t* skipped_updates) {
  if (!e || !e->ws) return fail("plb_status: engine not bound");
  unsigned int v[3] = {0, 0, 0};
  HIPTRY(hipDeviceSynchronize());
  HIPTRY(hipMemcpy(v, e->at<unsigned int>(e->o_lnerr), sizeof(v), hipMemcpyDeviceToHost));
  if (ln_exchange_timeouts) *ln_exchange_timeouts = (int32_t)v[0];
  if (skipped_updates) *skipped_updates = (int32_t)v[1];
  if (v[0]) {
    // the token head counts its own AdamW steps on the host (tok_steps): take back the ones the device left out (word 2)
    e->tok_steps = e->tok_steps > (int)v[2] ? e->tok_steps - (int)v[2] : 0;
    // Reported once, then gone: a producer whose store landed after its consumer had given up leaves a tagged granule
    // that the next launch would take for a fresh one, so the exchange buffer is zeroed again (the device is idle
    // here) together with the error word and its host mirror. The next step starts clean.
    HIPTRY(hipMemset(e->at<char>(e->o_lnx), 0, (size_t)e->lnx_bytes));
    HIPTRY(hipMemset(e->at<char>(e->o_lnerr), 0, 256));
    HIPTRY(hipDeviceSynchronize());
    if (e->host_err) *(volatile unsigned int*)e->host_err = 0;
  }
  return 0;
}
extern "C" int plb_status(PlbEngine* e, int32_t* ln_exchange_timeouts) { return plb_status_ex(e, ln_exchange_timeouts, nullptr); }

extern "C" int plb_poll_status(const PlbEngine* e, int32_t* ln_exchange_timeouts) {
  if (!e || !e->ws || !e->host_err) return fail("plb_poll_status: engine not bound");
  if (ln_exchange_timeouts) *ln_exchange_timeouts = (int32_t)*(volatile const unsigned int*)e->host_err;
  return 0;
}

// A host that exchanges the gradients ITSELF (torch.distributed fallback, a foreign communicator) must let the health word
// travel with them: export after the loss call, sum the float over the ranks, import before plb_adamw_step.
extern "C" int plb_status_export(PlbEngine* e, float* out, void* stream) {
  if (!e || !e->ws || !out) return fail("plb_status_export: bad argument");
  TRY(plb_launch_status_export(e->at<unsigned int>(e->o_lnerr), out, (hipStream_t)stream));
  return 0;
}
extern "C" int plb_status_import(PlbEngine* e, const float* summed, void* stream) {
  if (!e || !e->ws || !summed) return fail("plb_status_import: bad argument");
  TRY(plb_launch_step_status(e->at<unsigned int>(e->o_lnerr), e->last_loss, e->host_err_dev, summed, (hipStream_t)stream));
  return 0;
}

// ---- debug: happens-before audit, exchange trace -------------------------------------------------------------------------
extern "C" int plb_debug_hb_audit(PlbEngine* e, int32_t on, int32_t break_wait) {
  if (!e) return fail("plb_debug_hb_audit: null engine");
  e->hb = HbAudit();
  e->hb.on = on != 0;
  e->hb.break_wait = break_wait;
  return 0;
}
extern "C" int plb_debug_hb_report(const PlbEngine* e, int64_t* checks, int32_t* violations, char* first, int32_t first_bytes) {
  if (!e) return fail("plb_debug_hb_report: null engine");
  if (checks) *checks = e->hb.checks;
  if (violations) *violations = e->hb.violations;
  if (first && first_bytes > 0) snprintf(first, (size_t)first_bytes, "%s", e->hb.first.c_str());
  return 0;
}
extern "C" int plb_comm_trace(PlbEngine* e, int32_t on) {
  if (!e) return fail("plb_comm_trace: null engine");
  e->trace_on = on != 0;
  return 0;
}
// Timing of the last loss call's pieces, in milliseconds since the call's first launch: when the piece was released (the
// launch that completed its range had finished) and when its all-reduce had finished; tail_ms[2] = begin / end of the tail
// of weight-gradient GEMMs on the caller's stream. Synchronises on the events. Returns the number of pieces in *n.
extern "C" int plb_comm_trace_read(PlbEngine* e, int32_t max_pieces, int32_t* n, int64_t* begin, int64_t* end, float* released_ms,
                                   float* done_ms, float* tail_ms) {
  if (!e || !n) return fail("plb_comm_trace_read: bad argument");
  *n = 0;
  if (!e->tr_call0) return 0;
  if (tail_ms) { tail_ms[0] = tail_ms[1] = 0.f; }
  if (tail_ms && e->tr_tail_valid) {
    HIPTRY(hipEventSynchronize(e->tr_tail1));
    HIPTRY(hipEventElapsedTime(&tail_ms[0], e->tr_call0, e->tr_tail0));
    HIPTRY(hipEventElapsedTime(&tail_ms[1], e->tr_call0, e->tr_tail1));
  }
  for (auto& t : e->trace) {
    if (*n >= max_pieces) break;
    HIPTRY(hipEventSynchronize(t.done));
    if (begin) begin[*n] = t.a;
    if (end) end[*n] = t.b;
    if (released_ms) HIPTRY(hipEventElapsedTime(&released_ms[*n], e->tr_call0, t.released));
    if (done_ms) HIPTRY(hipEventElapsedTime(&done_ms[*n], e->tr_call0, t.done));
    *n += 1;
  }
  return 0;
}

extern "C" int plb_last_application_rows(const PlbEngine* e, int64_t* rows, int64_t* of) {
  if (!e) return fail("plb_last_application_rows: null engine");
  if (rows) *rows = e->last_call_rows[0];
  if (of) *of = e->last_call_rows[1];
  return 0;
}

extern "C" int plb_comm_pieces(const PlbEngine* e, int32_t* collectives, int64_t* floats) {
  if (!e) return fail("plb_comm_pieces: null engine");
  if (collectives) *collectives = e->piece_count;
  if (floats) *floats = e->piece_floats;
  return 0;
}

extern "C" int plb_set_grad_overlap(PlbEngine* e, int32_t overlap) {
  if (!e) return fail("plb_set_grad_overlap: null engine");
  e->overlap = overlap != 0;
  return 0;
}

extern "C" int plb_broadcast_params(PlbEngine* e, int32_t root, void* stream) {
  if (!e || !e->ws) return fail("plb_broadcast_params: engine not bound");
  if (!e->comm) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int rc = g_rccl.Broadcast(e->params, e->params, (size_t)e->ptotal, kNcclFloat32, root, e->comm, s);
  if (rc != kNcclSuccess) return fail("ncclBroadcast: %s", g_rccl.GetErrorString(rc));
  return plb_sync_weights(e, stream);
}

extern "C" int plb_allreduce_grads(PlbEngine* e, void* stream) {
  if (!e || !e->ws) return fail("plb_allreduce_grads: engine not bound");
  if (!e->comm) return 0;
  if (!e->grads) return fail("plb_allreduce_grads: no gradient buffer bound");
  hipStream_t s = (hipStream_t)stream;
  if (e->comm_pending) {  // the loss call issued the pieces: join them
    HIPTRY(ev_wait(e, s, e->ev_comm_done));
    e->comm_pending = false;
    return 0;
  }
  if (e->grads_reduced) return 0;
  HB_W(s, e->grads, e->ptotal * 4, "in-stream all-reduce of the gradient buffer");
  int rc = g_rccl.AllReduce(e->grads, e->grads, (size_t)e->ptrain, kNcclFloat32, kNcclSum, e->comm, s);
  e->piece_count = 1;
  e->piece_floats = e->ptrain;
  if (rc == kNcclSuccess && e->tok_grads_live) {
    const int64_t o = e->poff[PLB_TOK_W];
    rc = g_rccl.AllReduce(e->grads + o, e->grads + o, (size_t)(e->ptotal - o), kNcclFloat32, kNcclSum, e->comm, s);
    e->piece_count = 2;
    e->piece_floats += e->ptotal - o;
  }
  if (rc != kNcclSuccess) return fail("ncclAllReduce: %s", g_rccl.GetErrorString(rc));
  e->grads_reduced = true;
  return 0;
}

extern "C" int32_t plb_token_head_steps(const PlbEngine* e) { return e ? e->tok_steps : -1; }
extern "C" int plb_set_token_head_steps(PlbEngine* e, int32_t steps) {
  if (!e || steps < 0) return fail("plb_set_token_head_steps: bad argument");
  e->tok_steps = steps;
  return 0;
}

extern "C" int plb_apply_mask(const int64_t* ids, const int32_t* sample_off, const int32_t* word_off,
                              const int32_t* word_begin, const int32_t* word_len, const int8_t* action, const int64_t* repl,
                              const int64_t* word_token, int64_t sep_token, const int32_t* crop_start, int32_t B, int32_t S,
                              int32_t mask_id, int64_t* labels, int64_t* masked, int64_t* tokens, int32_t* lengths_out,
                              int32_t* idx_offsets, int32_t* idx_flat, int32_t* scratch, void* stream) {
  if (!ids || !sample_off || !word_off || !word_begin || !word_len || !action || !repl || !crop_start || !labels || !masked ||
      !lengths_out || !idx_offsets || !idx_flat || !scratch)
    return fail("plb_apply_mask: null argument");
  if ((tokens != nullptr) != (word_token != nullptr)) return fail("plb_apply_mask: tokens and word_token go together");
  if (S < 1 || S > 1024 || B < 1 || B > 1024) return fail("plb_apply_mask: needs 1 <= S <= 1024, 1 <= B <= 1024");
  PlbApplyMask m;
  memset(&m, 0, sizeof(m));
  m.ids = ids; m.sample_off = sample_off; m.word_off = word_off; m.word_begin = word_begin; m.word_len = word_len;
  m.action = action; m.repl = repl; m.word_token = word_token; m.sep_token = sep_token; m.crop_start = crop_start;
  m.B = B; m.S = S; m.mask_id = mask_id;
  m.labels = labels; m.masked = masked; m.tokens = tokens; m.lengths_out = lengths_out;
  m.counts = scratch; m.idx_padded = scratch + B; m.offsets = idx_offsets; m.flat = idx_flat;
  TRY(plb_launch_apply_mask(&m, (hipStream_t)stream));
  return 0;
}

extern "C" int plb_mask_batch(const int64_t* labels, const int32_t* lengths, int32_t B, int32_t S, uint64_t seed,
                              uint32_t step, float word_pred_prob, float phoneme_mask_prob, float replace_prob,
                              int32_t mask_id, int32_t sep_id, int64_t* masked, int32_t* idx_offsets, int32_t* idx_flat,
                              int32_t* scratch, void* stream) {
  if (!labels || !masked || !idx_offsets || !idx_flat || !scratch) return fail("plb_mask_batch: null argument");
  if (S < 1 || S > 512 || B < 1 || B > 1024) return fail("plb_mask_batch: needs 1 <= S <= 512, 1 <= B <= 1024");
  PlbMask m;
  memset(&m, 0, sizeof(m));
  m.labels = labels; m.lengths = lengths; m.B = B; m.S = S; m.seed = seed; m.step = step;
  m.word_pred_prob = word_pred_prob; m.mask_prob = phoneme_mask_prob; m.replace_prob = replace_prob;
  m.mask_id = mask_id; m.sep_id = sep_id;
  m.masked = masked; m.counts = scratch; m.idx_padded = scratch + B; m.offsets = idx_offsets; m.flat = idx_flat;
  TRY(plb_launch_mask(&m, (hipStream_t)stream));
  return 0;
}

extern "C" int plb_adamw_step(PlbEngine* e, double lr, double beta1, double beta2, double eps, double weight_decay,
                              int32_t step, double grad_scale, void* stream) {
  if (!e || !e->ws || !e->grads || !e->m || !e->v) return fail("plb_adamw_step: optimizer buffers not bound");
  if (e->infer) return fail("plb_adamw_step: inference-only engine");
  if (step < 1) return fail("plb_adamw_step: step counts from 1");
  hipStream_t s = (hipStream_t)stream;
  if (e->comm_pending) {  // all-reduce pieces still in flight on the communication stream
    HIPTRY(ev_wait(e, s, e->ev_comm_done));
    e->comm_pending = false;
  }
  HB_R(s, e->grads, e->ptotal * 4, "AdamW (reads the gradient buffer)");
  TRY(plb_launch_adamw(e->params, e->grads, e->m, e->v, e->at<bf16_t>(e->o_wbf), (size_t)e->ptrain, lr, beta1, beta2, eps,
                       weight_decay, step, grad_scale, e->at<unsigned int>(e->o_lnerr), 1, s));
  if (e->tok_grads_live) {
    // token head: trained only by dual-head steps (no gradient, no update — as the pooler), with its OWN step count:
    // torch.optim.AdamW keeps one per parameter, so a head that starts training late gets its own bias correction
    const int64_t o = e->poff[PLB_TOK_W];
    e->tok_steps += 1;
    TRY(plb_launch_adamw(e->params + o, e->grads + o, e->m + o, e->v + o, e->at<bf16_t>(e->o_wbf) + o,
                         (size_t)(e->ptotal - o), lr, beta1, beta2, eps, weight_decay, e->tok_steps, grad_scale,
                         e->at<unsigned int>(e->o_lnerr), 2, s));
  }
  return sync_transposes(e, s, false);   // fp8 copies: delayed scaling from here on (the weights moved by one AdamW step)
}
