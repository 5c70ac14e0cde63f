/* plbert.h — C ABI of the MI355X-native PL-BERT pre-training hot path (libplbert_hip.so).
 *
 * The reference (Fadi987/PL-BERT) has no FFI: its boundary for this path is a Python nn.Module
 * contract.  Each entry point below names the reference interface it stands in
 * for; the Python host in plbert_amd/ (model.py, engine.py) binds them with ctypes and mirrors the
 * reference classes on top.  Plain pointers and sizes only: device pointers are raw HBM addresses
 * (any allocator — the Python host passes torch tensors' data_ptr()), `stream` is a hipStream_t
 * passed as void* (NULL = the legacy default stream).  Every call is asynchronous on `stream`
 * unless stated otherwise.  Return value: 0 = ok, non-zero = error, text via plb_last_error().
 * One engine serves one GPU and one host thread at a time (the reference runs one process per
 * GPU with a single-threaded step loop, train.py:350-357).
 */
#ifndef PLBERT_H
#define PLBERT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PlbEngine PlbEngine;

/* The AlbertConfig subset of the path (train.py:263-265; configs/config.yml:32-39; HF defaults
 * configuration_albert.py:56-75) plus the head sizes (model.py:6,20) and workspace capacity. */
typedef struct {
  int32_t vocab_size;              /* len(symbols) = 188 (train.py:263) */
  int32_t embedding_size;          /* 128 */
  int32_t hidden_size;             /* 768 */
  int32_t num_attention_heads;     /* 12; hidden_size / heads must be 64 */
  int32_t intermediate_size;       /* 2048 */
  int32_t num_hidden_layers;       /* 12 applications of the one shared layer */
  int32_t max_position_embeddings; /* 512 */
  int32_t type_vocab_size;         /* 2 */
  float layer_norm_eps;            /* 1e-12 */
  int32_t num_phonemes;            /* phoneme_predictor out features (model.py:10,24) */
  int32_t num_tokens;              /* token_predictor out features, 0 = PhonemeOnlyModel (model.py:11) */
  int32_t max_batch;               /* capacity: largest B accepted by the calls below */
  int32_t max_seq;                 /* capacity: largest S (<= max_position_embeddings) */
  int32_t inference_only;          /* 1: forward / loss-only use (README.md:91, train.py:288-304 validate): the workspace
                                      keeps ONE layer's activations instead of all of them, no gradient stash; the
                                      backward and AdamW entry points then fail. 0: training engine. */
} PlbConfig;

/* Parameter tensors in flat-buffer order; names follow the reference state_dict.
 * query/key/value weights (and biases) are adjacent so they form the fused [3H,H] QKV operand. */
enum PlbParam {
  PLB_WORD_EMB = 0,   /* encoder.embeddings.word_embeddings.weight        [V,E]  */
  PLB_POS_EMB,        /* encoder.embeddings.position_embeddings.weight    [P,E]  */
  PLB_TYPE_EMB,       /* encoder.embeddings.token_type_embeddings.weight  [2,E]  */
  PLB_EMB_LN_W,       /* encoder.embeddings.LayerNorm.weight              [E]    */
  PLB_EMB_LN_B,       /* encoder.embeddings.LayerNorm.bias                [E]    */
  PLB_MAP_W,          /* encoder.encoder.embedding_hidden_mapping_in.weight [H,E] */
  PLB_MAP_B,          /* ...embedding_hidden_mapping_in.bias              [H]    */
  PLB_LN2_W,          /* <layer>.full_layer_layer_norm.weight             [H]    */
  PLB_LN2_B,          /* <layer>.full_layer_layer_norm.bias               [H]    */
  PLB_Q_W, PLB_K_W, PLB_V_W,   /* <layer>.attention.{query,key,value}.weight [H,H] each */
  PLB_Q_B, PLB_K_B, PLB_V_B,   /* <layer>.attention.{query,key,value}.bias   [H] each   */
  PLB_DENSE_W,        /* <layer>.attention.dense.weight                   [H,H]  */
  PLB_DENSE_B,        /* <layer>.attention.dense.bias                     [H]    */
  PLB_LN1_W,          /* <layer>.attention.LayerNorm.weight               [H]    */
  PLB_LN1_B,          /* <layer>.attention.LayerNorm.bias                 [H]    */
  PLB_FFN_W,          /* <layer>.ffn.weight                               [I,H]  */
  PLB_FFN_B,          /* <layer>.ffn.bias                                 [I]    */
  PLB_FFNO_W,         /* <layer>.ffn_output.weight                        [H,I]  */
  PLB_FFNO_B,         /* <layer>.ffn_output.bias                          [H]    */
  PLB_HEAD_W,         /* phoneme_predictor.weight                         [NP,H] */
  PLB_HEAD_B,         /* phoneme_predictor.bias                           [NP]   */
  /* --- everything below receives no gradient in the reference (train.py:383-390 never reads it) */
  PLB_POOL_W,         /* encoder.pooler.weight                            [H,H]  */
  PLB_POOL_B,         /* encoder.pooler.bias                              [H]    */
  PLB_TOK_W,          /* token_predictor.weight                           [NT,H] (size 0 when num_tokens = 0) */
  PLB_TOK_B,          /* token_predictor.bias                             [NT]   */
  PLB_NPARAM
};

/* Thread-local text of the last error returned by any call on this thread. */
const char* plb_last_error(void);

/* Stands in for: AlbertModel(AlbertConfig(...)) + PhonemeOnlyModel/MultiTaskModel construction
 * (train.py:263-270; model.py:6-11,20-24). Host-only; allocates no device memory. */
int plb_create(const PlbConfig* cfg, PlbEngine** out);
void plb_destroy(PlbEngine* e);

/* Flat parameter layout: offsets/sizes in floats for each PlbParam, the total float count and the
 * count of leading floats that receive gradients (AdamW range). Stands in for
 * nn.Module.parameters()/state_dict() (train.py:272,417). Arrays hold PLB_NPARAM entries. */
int plb_param_layout(const PlbEngine* e, int64_t* offsets, int64_t* sizes, int64_t* total, int64_t* trainable);

/* Bytes of device workspace the engine needs for its capacity (activations stashed for all
 * layers, bf16 weight copies, split-K slabs). The caller allocates it ZERO-FILLED. */
int64_t plb_workspace_bytes(const PlbEngine* e);

/* Borrow the caller's device buffers: fp32 params / grads / AdamW moments (each `total` floats,
 * laid out per plb_param_layout; grads and moments may be NULL for inference-only use) and the
 * workspace. Stands in for module.to(device) / accelerator.prepare (train.py:160-162). */
int plb_bind(PlbEngine* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* workspace,
             int64_t workspace_bytes);

/* Refresh the bf16 (and transposed) compute copies from the fp32 parameters. Call after the
 * parameters were written from outside (load_state_dict, train.py:100; broadcast at start-up).
 * plb_adamw_step does this itself. */
int plb_sync_weights(PlbEngine* e, void* stream);

/* Stands in for PhonemeOnlyModel.forward / MultiTaskModel.forward / AlbertModel.forward
 * (model.py:13-18,26-30; train.py:386; README.md:91): ids int64 [B,S]; lengths int32 [B] = number
 * of valid (attention_mask == 1) leading tokens per sample, or NULL for no padding. Any of the
 * outputs may be NULL: hidden fp32 [B,S,H] (.last_hidden_state), phoneme_logits fp32 [B,S,NP],
 * token_logits fp32 [B,S,NT]. */
int plb_forward(PlbEngine* e, const int64_t* ids, const int32_t* lengths, int32_t B, int32_t S, float* hidden,
                float* phoneme_logits, float* token_logits, void* stream);

/* AlbertModel's pooler_output (modeling_albert.py:403; computed by the reference, never used by its loss):
 * pooled[b,:] = tanh(pooler.weight · hidden[b,0,:] + pooler.bias); hidden fp32 [B,S,H] (plb_forward's output),
 * pooled fp32 [B,H]. */
int plb_pooler(PlbEngine* e, const float* hidden, int32_t B, int32_t S, float* pooled, void* stream);

/* Stands in for process_batch + calculate_phoneme_loss + accelerator.backward (train.py:381-390,
 * 107-131, 356): masked_ids/labels int64 [B,S]; lengths int32 [B]; the per-sample masked index
 * lists as CSR (idx_offsets int32 [B+1], idx_flat int32 [n_masked], positions < lengths[b], unique
 * within a sample). Writes the scalar loss (fp32, device) and the gradient of every trainable
 * parameter into the bound grads buffer (overwritten, not accumulated). n_masked == 0 reproduces
 * the reference's zero-loss fallback: loss 0 and all gradients 0. */
int plb_loss_fwd_bwd(PlbEngine* e, const int64_t* masked_ids, const int64_t* labels, const int32_t* lengths,
                     const int32_t* idx_offsets, const int32_t* idx_flat, int32_t n_masked, int32_t B, int32_t S,
                     float* loss, void* stream);

/* Stands in for process_batch under torch.no_grad() — validate() (train.py:288-304): forward + the same loss as
 * plb_loss_fwd_bwd (or plb_loss_fwd_bwd_dual when token_ids != NULL; loss_parts optional), NO backward: the bound
 * gradient buffer is not touched, nothing is stashed. Works on training and inference-only engines. */
int plb_loss_fwd(PlbEngine* e, const int64_t* masked_ids, const int64_t* labels, const int64_t* token_ids,
                 const int32_t* lengths, const int32_t* idx_offsets, const int32_t* idx_flat, int32_t n_masked, int32_t B,
                 int32_t S, float* loss, float* loss_parts, void* stream);

/* Dual-head training step for MultiTaskModel (model.py:5-18: phoneme_predictor + token_predictor) fed by the
 * 4-tuple Collater batch (dataloader.py:200-223: token_ids, labels, masked, lengths, indices). The reference
 * defines the head and the batch but trains PhonemeOnlyModel only (train.py:266-270); the token loss here is
 * upstream PL-BERT's: per-sample CrossEntropyLoss (mean) of token_pred[b, :len_b] against token_ids[b, :len_b],
 * averaged over the B samples. loss = phoneme loss (as plb_loss_fwd_bwd) + token loss; loss_parts (optional)
 * receives the two terms. token_ids int64 [B,S] in [0, num_tokens). Gradients of token_predictor.{weight,bias}
 * are written as well and the next plb_adamw_step updates them; after a plb_loss_fwd_bwd call the token head
 * has no gradient and is left alone, like the pooler. n_masked == 0 is allowed (phoneme term 0). */
int plb_loss_fwd_bwd_dual(PlbEngine* e, const int64_t* masked_ids, const int64_t* labels, const int64_t* token_ids,
                          const int32_t* lengths, const int32_t* idx_offsets, const int32_t* idx_flat, int32_t n_masked,
                          int32_t B, int32_t S, float* loss, float* loss_parts, void* stream);

/* Stands in for torch.optim.AdamW.step (train.py:272,357): decoupled weight decay on every
 * trainable parameter, bias-corrected moments; `step` counts from 1; gradients are multiplied by
 * grad_scale first (1/world_size after a sum all-reduce). Also refreshes the bf16 copies. Hyper-parameters are
 * doubles, as torch holds them: the update's scalars (1 - beta2, lr / bias_correction1, ...) are formed in double and
 * rounded to fp32 once, which is what makes the result agree with torch.optim.AdamW to the last bits. */
int plb_adamw_step(PlbEngine* e, double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step,
                   double grad_scale, void* stream);

/* fp8 mode (BASELINE.json configs[4]): EVERY projection GEMM of the shared layer runs on OCP fp8 operands with fp32
 * accumulation through the block-scaled MFMA with unit block scales (twice the bf16 MFMA rate) — forward: QKV, dense
 * (+ LayerNorm 1), FFN up (+ gelu_new), FFN output (+ LayerNorm 2); backward: the four dX GEMMs (two of them carrying a
 * LayerNorm backward, one the gelu backward) and the four weight-gradient GEMMs over all applications. e4m3 weights and
 * activations, e5m2 gradients. The 1-byte images are written by the launches that produce the tensors (fused epilogues,
 * attention kernels), one per application of the layer; attention itself, the head, LayerNorm / softmax statistics, the
 * loss and the optimizer stay bf16 / fp32. Delayed scaling, one scale per operand SITE shared by the L applications: a
 * tensor is quantised with 448 / the maximum its site showed in the previous call (gradients: 28672 / the maximum, one
 * binade of headroom); the first call after switching the mode on — and a training call whose gradient sites have not
 * been seen yet — runs in bf16 and only records the maxima. hidden_size 768 or 1024; calls whose GEMM shapes have no
 * pipeline-tile form run in bf16. The reference has no fp8 path (configs/config.yml:15: fp16 autocast): parity is against
 * this library's own bf16 path (tests/test_gpu_fp8.py: loss within 2e-2, whole-gradient relative L2 0.11) and against a
 * CPU restatement of exactly these semantics on top of the pinned oracle (oracle/fp8_np.py). */
int plb_set_fp8(PlbEngine* e, int32_t on, void* stream);
int plb_fp8_state(const PlbEngine* e, int32_t* enabled, int32_t* calibrated);
/* Delayed scaling and its limits. Every operand site quantises a call's values with the scale the PREVIOUS calls' maxima
 * gave — the LARGEST maximum of the last four calls: activations (e4m3) map it to 448, gradients (e5m2) to 28672 = half the
 * format's range. A call whose values exceed everything seen in the last four calls (gradients: 2x that) has the excess
 * CLAMPED in every GEMM operand of that site — silently as far as the loss goes. plb_fp8_stats makes it visible: per site
 * (order: x, a, gelu(u), context; dpre2, dU, dpre1, dQKV), the number of calls since plb_set_fp8 in which values were
 * clamped by more than the top value's rounding step (true maximum x scale > 1.0625 x format maximum) and the worst such
 * overshoot. Synchronises `stream`. No reference counterpart (the reference has no fp8 path). */
int plb_fp8_stats(PlbEngine* e, float clamped_calls[8], float worst_overshoot[8], void* stream);

/* The token head (trained by dual-head steps only) keeps its own AdamW step count, as torch keeps one per parameter
 * (train.py:417-421 saves it in 'optimizer'). Read / restore it around checkpoints. */
int32_t plb_token_head_steps(const PlbEngine* e);
int plb_set_token_head_steps(PlbEngine* e, int32_t steps);

/* ---- data-parallel exchange step (train.py:218-221,160-162,356: accelerate -> DDP -> NCCL; here RCCL over xGMI) ----
 * One process per GPU, one engine per process. Rank 0 calls plb_comm_unique_id and hands the 128 bytes to every rank
 * over any host channel (the Python host: torch.distributed's store); then every rank calls plb_comm_init (collective,
 * blocks until all ranks arrived). The library resolves RCCL at run time (the librccl.so.1 already mapped in the
 * process, else the system one; PLBERT_RCCL_LIB overrides) - it is not linked against it. */
#define PLB_COMM_ID_BYTES 128
int plb_comm_unique_id(uint8_t id[PLB_COMM_ID_BYTES]);
int plb_comm_init(PlbEngine* e, const uint8_t id[PLB_COMM_ID_BYTES], int32_t rank, int32_t world);
int plb_comm_destroy(PlbEngine* e);
int plb_comm_info(const PlbEngine* e, int32_t* rank, int32_t* world, int32_t* rccl_version);
/* Health of the in-launch hand-offs between the column tiles of the LayerNorm-in-GEMM launches (csrc/gemm_ln.hip). Their
 * wait is bounded so that a launch whose tiles were not co-resident long enough ENDS instead of hanging the device; the
 * launch then raises the engine's error word, and by construction (no host round trip anywhere):
 *   - the loss that call returns is NaN, and the word is mirrored into pinned host memory by the call's last launch;
 *   - plb_adamw_step leaves parameters, moments and compute copies untouched while the word is set;
 *   - the word stays set — every later step is skipped the same way — until plb_status has reported it.
 * plb_poll_status: NO synchronisation; the count as of the last loss call that has COMPLETED on the device (read it
 *   wherever the host has just read a loss back, or one step late at the top of the next step).
 * plb_status / plb_status_ex: SYNCHRONISES the device and returns the count since its previous report; a non-zero report also re-zeroes
 *   the exchange buffer and the word (a producer's store that landed after its consumer gave up would otherwise look
 *   fresh to the next launch), so the step after a reported failure starts clean. 0 in every run so far.
 * DATA-PARALLEL RUNS: the word is agreed between the ranks inside every plb_loss_fwd_bwd[_dual] call of an engine with a
 *   communicator — one float per rank, summed by a one-element all-reduce issued after the last launch that can raise
 *   it (overlap on: on the communication stream, between the head piece and the first weight's; overlap off: in
 *   `stream`) and merged by the call's last launch. A time-out on ONE rank therefore gives EVERY rank a NaN loss, a
 *   skipped update and a non-zero count (the sum over the ranks): the replicas stay bit-identical, nobody applies the
 *   poisoned sum. The status all-reduce is not counted by plb_comm_pieces. A host that exchanges gradients by other
 *   means uses plb_status_export / plb_status_import (below) at the same point. plb_loss_fwd (validation) issues no
 *   collective: a word raised there is sticky and travels with the next training call.
 * No reference counterpart (the reference's LayerNorm is torch's kernel); the Python host raises HandoffTimeout. */
int plb_status(PlbEngine* e, int32_t* ln_exchange_timeouts);
/* plb_status + the number of plb_adamw_step calls the device left out since the word was raised (a host that counts
 * optimizer steps for the bias correction rewinds its count by it); either pointer may be NULL. */
int plb_status_ex(PlbEngine* e, int32_t* ln_exchange_timeouts, int32_t* skipped_updates);
int plb_poll_status(const PlbEngine* e, int32_t* ln_exchange_timeouts);
/* For a host that runs the gradient exchange itself (torch.distributed, a foreign communicator): plb_status_export writes
 * this rank's count as one float to `out` (device) behind the loss call in `stream`; the host sums it over the ranks;
 * plb_status_import merges the sum into the word, mirrors it to the host and turns the last loss call's loss into NaN —
 * before plb_adamw_step (the loss buffer handed to that loss call must still be valid: the import writes the NaN there).
 * (With the engine's own communicator both happen inside plb_loss_fwd_bwd.) */
int plb_status_export(PlbEngine* e, float* out, void* stream);
int plb_status_import(PlbEngine* e, const float* summed, void* stream);
/* A phoneme-only loss call evaluates the part of its LAST shared-layer application that lies behind the attention
 * (dense + LayerNorm, FFN, LayerNorm) on the masked rows alone, forward and backward: the reference computes every row
 * and then reads the masked ones (train.py:107-131: pred[b, :len_b][idx_b]), rows only meet inside attention, so the other
 * rows of that part reach neither the loss nor — their output gradient being exactly zero — any gradient. Same results,
 * ~5 % less arithmetic at 13 % masked positions. Reports what the last loss call did: rows = token rows that part ran on
 * (padded to 128), of = the call's padded token count (rows == of: every row — a dual-head call, an fp8 call, more than
 * half of the positions masked, or PLBERT_PRUNE_LAST=0). */
int plb_last_application_rows(const PlbEngine* e, int64_t* rows, int64_t* of);
/* What the last training step exchanged: the number of collectives it issued (10 pieces for the reference's phoneme-only
 * step with overlap on, 1 with overlap off; one more after a dual-head step) and the floats they covered. The reference
 * has no counterpart (DDP's bucket count is internal to torch, train.py:218-221); a caller logs it to see which form of
 * the exchange actually ran. */
int plb_comm_pieces(const PlbEngine* e, int32_t* collectives, int64_t* floats);
/* DDP's start-up broadcast of rank `root`'s parameters; refreshes the bf16 copies. */
int plb_broadcast_params(PlbEngine* e, int32_t root, void* stream);
/* overlap = 1 (default): plb_loss_fwd_bwd[_dual] itself issues the gradient all-reduce, in contiguous pieces of the
 * flat buffer on the engine's communication stream as each piece becomes final (head gradients at the start of the
 * backward; the shared layer's weight gradients one by one, each while the next weight-gradient GEMM runs), and
 * plb_allreduce_grads only joins it. overlap = 0: plb_allreduce_grads performs ONE all-reduce in `stream`. */
int plb_set_grad_overlap(PlbEngine* e, int32_t overlap);
/* Sum all-reduce of every gradient the last loss call produced (the trainable range; plus the token head after a
 * dual-head step) over the ranks of the communicator; plb_adamw_step's grad_scale = 1/world turns it into DDP's mean.
 * After the call returns the reduced gradients are ordered before later work on `stream`. No communicator: no-op. */
int plb_allreduce_grads(PlbEngine* e, void* stream);

/* Bit-exact application of the reference's word-level masking (dataloader.py:59-137) on the device. The HOST draws the
 * reference's random streams in the reference's order (plbert_amd/data.py: MaskedPhonemeDataset.decisions) and hands
 * over, per sample, the UNCROPPED phoneme ids with a separator after each word, the word boundaries and one decision
 * per word; the device writes labels / masked ids (cropped to max_seq_length at crop_start, zero padded to S — the
 * collater's pad, dataloader.py:276-297) and the masked index list re-based to the crop (CSR), exactly as
 * __getitem__ + the collater produce them. Integer work only: results are bit-identical to the reference's.
 *   ids int64 [n_total]: all samples' uncropped ids back to back; sample_off int32 [B+1]: offsets into ids;
 *   word_off int32 [B+1]: offsets into word_begin / word_len / action / word_token (one entry per word: first position
 *   inside the sample, phoneme count, action 0 keep | 1 mask | 2 replace, grapheme-token id);
 *   repl int64 [n_total]: replacement ids (read at the positions of action-2 words); crop_start int32 [B];
 *   word_token + tokens (both NULL, or both given): the 4-tuple Collater's token_ids row, every phoneme of a word
 *   carrying the word's token id and separators sep_token (dataloader.py:66-71).
 * Outputs: labels / masked / tokens int64 [B,S]; lengths_out int32 [B] = min(len, S); idx_offsets int32 [B+1];
 * idx_flat int32 [capacity B*S]; scratch int32 [B + B*S]. S <= 1024, B <= 1024. */
int plb_apply_mask(const int64_t* ids, const int32_t* sample_off, const int32_t* word_off, const int32_t* word_begin,
                   const int32_t* word_len, const int8_t* action, const int64_t* repl, const int64_t* word_token,
                   int64_t sep_token, const int32_t* crop_start, int32_t B, int32_t S, int32_t mask_id, int64_t* labels,
                   int64_t* masked, int64_t* tokens, int32_t* lengths_out, int32_t* idx_offsets, int32_t* idx_flat,
                   int32_t* scratch, void* stream);

/* Device-side fast mode of the word-level masking that MaskedPhonemeDataset does on the host
 * (dataloader.py:83-108): same decision tree and probabilities (select a word with word_pred_prob;
 * then mask with phoneme_mask_prob / replace from the sample's own phonemes with replace_prob / keep),
 * separators never masked or indexed, but counter-based Philox randomness keyed by (seed, step, sample,
 * word) instead of the reference's global NumPy/Python streams — distribution-matched, NOT bit-exact
 * (the bit-exact path is the host one in plbert_amd/data.py). Needs no engine.
 * labels int64 [B,S] (separator id sep_id between words, positions >= lengths[b] ignored); outputs:
 * masked int64 [B,S], idx_offsets int32 [B+1], idx_flat int32 [up to B*S], scratch int32 [B + B*S].
 * S <= 512, B <= 1024. The total count is idx_offsets[B] (read it back before plb_loss_fwd_bwd). */
int plb_mask_batch(const int64_t* labels, const int32_t* lengths, int32_t B, int32_t S, uint64_t seed, uint32_t step,
                   float word_pred_prob, float phoneme_mask_prob, float replace_prob, int32_t mask_id, int32_t sep_id,
                   int64_t* masked, int32_t* idx_offsets, int32_t* idx_flat, int32_t* scratch, void* stream);

/* Measurement aid (no reference counterpart; the reference has no profiling): when
 * enabled, every kernel launch is bracketed by two HIP events on its stream. plb_profile_read waits
 * for them and returns, per kernel class, total milliseconds, launch count and the algorithmic
 * flops / bytes of those launches (arrays of plb_profile_num_classes() entries), then clears. */
void plb_profile_enable(int on);
int plb_profile_num_classes(void);
const char* plb_profile_class_name(int cls);
int plb_profile_read(double* ms, int64_t* launches, double* flops, double* bytes);

/* ---- test and tuning hooks (exported by the same library; NOT part of the drop-in surface, no reference counterpart).
 * They change process-wide state of the launchers and exist for tests/ and tools/ only:
 *   plb_debug_skip_piece(i)       the next loss call leaves out its i-th all-reduce piece, as a forgotten tensor would;
 *                                 the call must fail ("gradient exchange covered ..."), tests/test_gpu_comm_fake_rccl.py
 *   plb_debug_ln_fault(mode, n)   the next n fused LayerNorm launches run with a broken hand-off (1: one column tile
 *                                 never publishes; 2: consumed granules stay tagged), tests/test_gpu_handoff_fault.py
 *   plb_debug_hb_audit(e, on, k)  happens-before audit of the backward's three streams: a host-side
 *                                 vector-clock model of every event record / stream wait the engine issues, checked
 *                                 against the cross-stream buffer accesses of each launch; a violation fails the loss
 *                                 call. k >= 0: the MODEL forgets the k-th wait of the next call (the audit must then
 *                                 report it; the HIP call is still made). Also PLBERT_HB_AUDIT=1. plb_debug_hb_report
 *                                 returns the checks made, the violations and the first one's text.
 *   plb_comm_trace / plb_comm_trace_read
 *                                 timing events around every all-reduce piece of the last loss call: when the piece was
 *                                 released, when its collective had finished, begin / end of the weight-gradient tail
 *                                 (bench.py --gpus N prints them: RCCL-vs-GEMM contention readable from one line)
 *   plb_set_gemm_nt_tile / plb_set_gemm_nt_prefetch / plb_set_attn_bwd_fused
 *                                 force a tile, a K-loop form or the attention-backward form for the launches that
 *                                 follow (0 / -1 / -1 restore the per-shape policy; attention backward: 1 single-kernel
 *                                 form, 0 two kernels, 2 policy + split by sample), tests/test_gpu_kernels.py, tools/ */
void plb_debug_skip_piece(int index);
void plb_debug_ln_fault(int mode, int launches);
int plb_debug_hb_audit(PlbEngine* e, int32_t on, int32_t break_wait);
int plb_debug_hb_report(const PlbEngine* e, int64_t* checks, int32_t* violations, char* first, int32_t first_bytes);
int plb_comm_trace(PlbEngine* e, int32_t on);
int plb_comm_trace_read(PlbEngine* e, int32_t max_pieces, int32_t* n, int64_t* begin, int64_t* end, float* released_ms,
                        float* done_ms, float* tail_ms);
void plb_set_gemm_nt_tile(int tile);
void plb_set_gemm_nt_prefetch(int on);
void plb_set_attn_bwd_fused(int on);
/* 0: every loss call evaluates every row of the last application (plb_last_application_rows), 1: masked rows only where
 * the call qualifies, -1: PLBERT_PRUNE_LAST's choice (default on). tests/test_gpu_engine.py compares the two. */
void plb_set_prune_last(int on);

#ifdef __cplusplus
}
#endif
#endif /* PLBERT_H */
