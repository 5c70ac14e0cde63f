// Device-side word-level masking: the fast mode of the masking path.
//
// The reference masks on the host, one Python string operation per word, from NumPy's and Python's
// global generators (dataloader.py:83-108) — that path is reproduced bit-exactly on the host by
// plbert_amd/data.py. This kernel is the distribution-matched device version for when the input
// pipeline must keep up with the GPUs: same decision tree and probabilities, counter-based
// Philox4x32-10 randomness keyed by (seed, step, sample, word / position), so a batch is a pure
// function of its inputs (reproducible, order-independent), but NOT the reference's bit stream.
//
//   per word (maximal run of non-separator tokens inside the sample's length):
//     u1 < word_pred_prob ?  ->  u2 < phoneme_mask_prob            : every phoneme -> MASK (185)
//                                u2 < phoneme_mask_prob+replace_prob: every phoneme -> a uniformly drawn
//                                                                     phoneme of the same sample (the pool
//                                                                     excludes separators, dataloader.py:37,94)
//                                otherwise                          : unchanged, and NOT indexed
//     masked_index = positions of the phonemes of masked / replaced words, ascending; separators are
//     never indexed (dataloader.py:101-104).
#include "common.h"
#include "plbert_kernels.h"

namespace {

struct U4 { uint32_t x, y, z, w; };

DEVI U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
DEVI float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1), 24 bits

// One workgroup of 512 threads per sample, S <= 512: thread i owns position i.
__global__ __launch_bounds__(512) void mask_words_kernel(PlbMask p) {
  __shared__ int wsep[8], wmod[8], wpool[8];
  __shared__ int pool[512];
  const int b = blockIdx.x, i = threadIdx.x, lane = i & 63, w = i >> 6;
  const int S = p.S;
  int len = p.lengths ? p.lengths[b] : S;
  len = len < 0 ? 0 : (len > S ? S : len);
  const bool in = i < len;
  const long long id = (i < S) ? p.labels[(size_t)b * S + i] : 0;
  const bool sep = in && id == p.sep_id;
  const bool ph = in && !sep;

  // exclusive counts of separators (-> word index) and of phonemes (-> pool rank) before position i
  const unsigned long long ms = __builtin_amdgcn_ballot_w64(sep), mp = __builtin_amdgcn_ballot_w64(ph);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (lane == 0) { wsep[w] = __builtin_popcountll(ms); wpool[w] = __builtin_popcountll(mp); }
  __syncthreads();
  int word = __builtin_popcountll(ms & below), rank = __builtin_popcountll(mp & below), npool = 0;
  for (int k = 0; k < 8; ++k) {
    if (k < w) { word += wsep[k]; rank += wpool[k]; }
    npool += wpool[k];
  }
  if (ph) pool[rank] = (int)id;
  __syncthreads();

  // word decision: identical for every phoneme of the word (same counter)
  bool modified = false;
  long long out = id;
  if (ph) {
    const U4 r = philox4x32_10(U4{(uint32_t)word, (uint32_t)b, (uint32_t)p.step, 0u}, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
    if (u01(r.x) < p.word_pred_prob) {
      const float u2 = u01(r.y);
      if (u2 < p.mask_prob) {
        out = p.mask_id;
        modified = true;
      } else if (u2 < p.mask_prob + p.replace_prob) {
        const U4 c = philox4x32_10(U4{(uint32_t)i, (uint32_t)b, (uint32_t)p.step, 1u}, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
        int pick = (int)(u01(c.x) * (float)npool);
        pick = pick < npool ? pick : npool - 1;
        out = pool[pick];
        modified = true;
      }
    }
  }
  if (i < S) p.masked[(size_t)b * S + i] = out;

  // ascending index list of modified positions
  const unsigned long long mm = __builtin_amdgcn_ballot_w64(modified);
  if (lane == 0) wmod[w] = __builtin_popcountll(mm);
  __syncthreads();
  int pos = __builtin_popcountll(mm & below), total = 0;
  for (int k = 0; k < 8; ++k) {
    if (k < w) pos += wmod[k];
    total += wmod[k];
  }
  if (modified) p.idx_padded[(size_t)b * S + pos] = i;
  if (i == 0) p.counts[b] = total;
}

// ---- bit-exact application of host-drawn decisions (dataloader.py:59-137) ------------------------------------------
// One workgroup per sample. Pass 1: every output position copies its uncropped source id (labels = masked = id,
// token = separator token; zero past the length). Pass 2: one thread per word rewrites the positions of its word
// that fall inside the crop window (mask id / replacement id; the word's token id) and flags the modified ones.
// Pass 3: the flags are compacted into the ascending index list (re-based to the crop by construction).
__global__ __launch_bounds__(256) void apply_mask_kernel(PlbApplyMask p) {
  __shared__ unsigned char flag[1024];
  __shared__ int wcnt[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int S = p.S;
  const int base = p.sample_off[b], n = p.sample_off[b + 1] - base;
  const int start = p.crop_start[b];
  int len = n - start;
  len = len > S ? S : (len < 0 ? 0 : len);
  for (int i = tid; i < S; i += 256) {
    const long long id = i < len ? p.ids[base + start + i] : 0;
    p.labels[(size_t)b * S + i] = id;
    p.masked[(size_t)b * S + i] = id;
    if (p.tokens) p.tokens[(size_t)b * S + i] = i < len ? p.sep_token : 0;
    flag[i] = 0;
  }
  if (tid == 0) p.lengths_out[b] = len;
  __syncthreads();
  const int w0 = p.word_off[b], nw = p.word_off[b + 1] - w0;
  for (int k = tid; k < nw; k += 256) {
    const int wb = p.word_begin[w0 + k], wl = p.word_len[w0 + k], act = p.action[w0 + k];
    const long long tk = p.word_token ? p.word_token[w0 + k] : 0;
    if (act == 0 && !p.tokens) continue;
    for (int q = 0; q < wl; ++q) {
      const int i = wb + q - start;
      if (i < 0 || i >= len) continue;
      if (act == 1) p.masked[(size_t)b * S + i] = p.mask_id;
      else if (act == 2) p.masked[(size_t)b * S + i] = p.repl[base + wb + q];
      if (act != 0) flag[i] = 1;
      if (p.tokens) p.tokens[(size_t)b * S + i] = tk;
    }
  }
  __syncthreads();
  int run = 0;  // modified positions before the current 256-position chunk
  for (int c0 = 0; c0 < S; c0 += 256) {
    const int i = c0 + tid;
    const bool mod = i < S && flag[i];
    const unsigned long long mm = __builtin_amdgcn_ballot_w64(mod);
    if (lane == 0) wcnt[w] = __builtin_popcountll(mm);
    __syncthreads();
    int pos = run + __builtin_popcountll(mm & ((1ull << lane) - 1ull)), tot = 0;
    for (int k = 0; k < 4; ++k) {
      if (k < w) pos += wcnt[k];
      tot += wcnt[k];
    }
    if (mod) p.idx_padded[(size_t)b * S + pos] = i;
    run += tot;
    __syncthreads();
  }
  if (tid == 0) p.counts[b] = run;
}

// offsets = exclusive scan of counts (B <= 1024); flat = concatenation of the per-sample lists.
__global__ __launch_bounds__(1024) void mask_compact_kernel(PlbMask p) {
  __shared__ int sc[1024];
  const int t = threadIdx.x;
  sc[t] = t < p.B ? p.counts[t] : 0;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = t >= o ? sc[t - o] : 0;
    __syncthreads();
    sc[t] += v;
    __syncthreads();
  }
  if (t < p.B) p.offsets[t + 1] = sc[t];
  if (t == 0) p.offsets[0] = 0;
  __syncthreads();
  for (int b = 0; b < p.B; ++b) {
    const int base = b ? sc[b - 1] : 0, n = sc[b] - base;
    for (int j = t; j < n; j += 1024) p.flat[base + j] = p.idx_padded[(size_t)b * p.S + j];
  }
}

}  // namespace

extern "C" int plb_launch_apply_mask(const PlbApplyMask* p, hipStream_t stream) {
  if (p->S < 1 || p->S > 1024 || p->B < 1 || p->B > 1024) return 1;
  hipLaunchKernelGGL(apply_mask_kernel, dim3(p->B), dim3(256), 0, stream, *p);
  PlbMask m = PlbMask();  // CSR compaction shared with the Philox fast mode
  m.B = p->B; m.S = p->S; m.counts = p->counts; m.idx_padded = p->idx_padded; m.offsets = p->offsets; m.flat = p->flat;
  hipLaunchKernelGGL(mask_compact_kernel, dim3(1), dim3(1024), 0, stream, m);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int plb_launch_mask(const PlbMask* p, hipStream_t stream) {
  if (p->S < 1 || p->S > 512 || p->B < 1 || p->B > 1024) return 1;
  hipLaunchKernelGGL(mask_words_kernel, dim3(p->B), dim3(512), 0, stream, *p);
  hipLaunchKernelGGL(mask_compact_kernel, dim3(1), dim3(1024), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
