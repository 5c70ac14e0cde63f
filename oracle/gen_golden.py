"""Capture golden vectors from the reference itself.  Runs ONLY in the build container.

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Imports the reference modules from /root/reference (read-only) and HuggingFace ``AlbertModel``
(the third-party arithmetic the reference calls), feeds them seeded inputs and deterministic
weights from ``plbert_amd.deterministic_state_dict``, and stores inputs + expected outputs as small
.npz fixtures.  Nothing here travels to the GPU box except the fixtures; the reference's Python
never leaves this container.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import os
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

torch.set_num_threads(8)

import accelerate  # noqa: E402,F401  (must be imported before the wandb stub: it probes for wandb)

sys.modules.setdefault("wandb", types.ModuleType("wandb"))  # train.py:11 imports it; never called here
sys.path.insert(0, REF)
import char_indexer as ref_char_indexer  # noqa: E402
import dataloader as ref_dataloader  # noqa: E402
import model as ref_model  # noqa: E402
import train as ref_train  # noqa: E402
from transformers import AlbertConfig, AlbertModel  # noqa: E402

import plbert_amd  # noqa: E402

DATASET_PARAMS = dict(word_separator=87, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1)


def obj_array(list_of_lists):
    a = np.empty(len(list_of_lists), dtype=object)
    for i, x in enumerate(list_of_lists):
        a[i] = np.asarray(x, dtype=np.int64)
    return a


def toy_docs():
    """Small documents over the real vocabulary (IPA + punctuation + an out-of-vocabulary char)."""
    rs = np.random.RandomState(7)
    sym = plbert_amd.symbols
    letters = sym[52:185]
    docs = []
    for n_words in (8, 5, 40, 1, 17, 120, 3, 64):
        words = []
        for _ in range(n_words):
            if rs.rand() < 0.12:
                words.append(sym[1 + rs.randint(0, 51)])            # a punctuation "word"
            else:
                words.append("".join(letters[i] for i in rs.randint(0, len(letters), size=rs.randint(1, 9))))
        docs.append(words)
    docs[1][2] = docs[1][2] + "¤"                               # '¤' is not in the table -> U
    return docs


def gen_masking():
    docs = toy_docs()
    out = {"symbols_codepoints": np.array([ord(c) for c in ref_char_indexer.symbols], dtype=np.int64)}
    data3 = [{"phonemes": d} for d in docs]
    rs = np.random.RandomState(11)
    data4 = [{"phonemes": d, "token_ids": rs.randint(100, 60000, size=len(d)).tolist()} for d in docs]
    for tag, msl in (("msl512", 512), ("msl32", 32)):
        # --- 3-tuple stream (use_token_ids=False) -------------------------------------------------
        np.random.seed(1)
        random.seed(1)
        ds = ref_dataloader.MaskedPhonemeDataset(data3, max_seq_length=msl, use_token_ids=False, **DATASET_PARAMS)
        order = [0, 1, 2, 3, 4, 5, 6, 7, 2, 5, 0, 7, 5, 5, 1, 4]
        items = [ds[i] for i in order]
        out[f"{tag}_order"] = np.array(order)
        out[f"{tag}_labels"] = obj_array([it[0].numpy() for it in items])
        out[f"{tag}_masked"] = obj_array([it[1].numpy() for it in items])
        out[f"{tag}_index"] = obj_array([it[2] for it in items])
        lab, msk, lens, idx = ref_dataloader.PhonemeOnlyCollater()(items[:8])
        out[f"{tag}_c3_labels"], out[f"{tag}_c3_masked"] = lab.numpy(), msk.numpy()
        out[f"{tag}_c3_lengths"] = np.array(lens)
        out[f"{tag}_c3_index"] = obj_array(idx)
        out[f"{tag}_c3_text_mask"] = ref_train.length_to_mask(torch.Tensor(lens)).numpy()
        # --- 4-tuple stream (use_token_ids=True) --------------------------------------------------
        np.random.seed(1)
        random.seed(1)
        ds4 = ref_dataloader.MaskedPhonemeDataset(data4, max_seq_length=msl, use_token_ids=True, **DATASET_PARAMS)
        items4 = [ds4[i] for i in order[:8]]
        tok, lab, msk, lens, idx = ref_dataloader.Collater()(items4)
        out[f"{tag}_c4_tokens"], out[f"{tag}_c4_labels"], out[f"{tag}_c4_masked"] = tok.numpy(), lab.numpy(), msk.numpy()
        out[f"{tag}_c4_lengths"] = np.array(lens)
        out[f"{tag}_c4_index"] = obj_array(idx)
    out["docs"] = np.array(["\x1f".join(d) for d in docs], dtype=object)
    out["token_ids"] = obj_array([r["token_ids"] for r in data4])
    # SURVEY.md §3.3 known answer: 2-doc toy set, 4th __getitem__ call, max_seq_length 16
    np.random.seed(1)
    random.seed(1)
    toy = [{"phonemes": ["aa", "bb", "cc", "dd", "ee", "ff", "gg", "hh"]}, {"phonemes": ["ab", "cd", "ef"]}]
    ds = ref_dataloader.MaskedPhonemeDataset(toy, max_seq_length=16, use_token_ids=False, **DATASET_PARAMS)
    calls = [1, 0, 1, 0]
    its = [ds[i] for i in calls]
    out["ka_calls"] = np.array(calls)
    out["ka_labels"] = obj_array([i[0].numpy() for i in its])
    out["ka_masked"] = obj_array([i[1].numpy() for i in its])
    out["ka_index"] = obj_array([i[2] for i in its])
    # train/val split order of build_dataloader (random.shuffle on the stdlib stream)
    np.random.seed(1)
    random.seed(1)
    big = [{"phonemes": ["ab", "cd"]} for _ in range(50)]
    tl, vl = ref_dataloader.build_dataloader(big, batch_size=4, device="cpu",
                                             dataset_config=dict(max_seq_length=512, **DATASET_PARAMS),
                                             use_token_ids=False)
    out["split_train_indices"] = np.array(tl.dataset.indices)
    out["split_val_indices"] = np.array(vl.dataset.indices)
    np.savez_compressed(os.path.join(OUT, "masking.npz"), **out, allow_pickle=True)
    print("masking.npz written")


def build_reference(cfg_kwargs, num_phonemes, num_tokens, sd, attn="eager"):
    hf_cfg = AlbertConfig(vocab_size=cfg_kwargs["vocab_size"], attn_implementation=attn,
                          **{k: v for k, v in cfg_kwargs.items() if k != "vocab_size"})
    enc = AlbertModel(hf_cfg)
    if num_tokens:
        m = ref_model.MultiTaskModel(enc, num_phonemes=num_phonemes, num_tokens=num_tokens,
                                     hidden_size=cfg_kwargs["hidden_size"])
    else:
        m = ref_model.PhonemeOnlyModel(enc, num_phonemes=num_phonemes, hidden_size=cfg_kwargs["hidden_size"])
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    return m


class _Acc:
    device = torch.device("cpu")


def ragged_batch(B, S, lengths, seed, vocab_hi=185):
    """Padded batch (sorted by length, zero padded) with explicit masked-index lists."""
    rs = np.random.RandomState(seed)
    labels = np.zeros((B, S), np.int64)
    masked = np.zeros((B, S), np.int64)
    idxs = []
    for b, L in enumerate(lengths):
        lab = rs.randint(1, vocab_hi, size=L)
        msk = lab.copy()
        n = max(1, L // 6)
        idx = sorted(rs.choice(L, size=n, replace=False).tolist())
        for i in idx[: max(1, (3 * n) // 4)]:
            msk[i] = 185
        labels[b, :L], masked[b, :L] = lab, msk
        idxs.append(idx)
    return labels, masked, list(lengths), idxs


def capture_model(tag, cfg_kwargs, num_phonemes, num_tokens, batch, seed, full, n_steps, lr=7e-5, init="deterministic"):
    """Run reference forward / loss / backward / AdamW and store what the tests compare.

    ``init`` names the weight generator both sides regenerate from ``seed``: "deterministic" (every tensor noisy, so
    bias paths are exercised) or "reference" (``plbert_amd.reference_init_state_dict`` = the reference's own
    initialisation, train.py:263-270 — what ``PLBertTrainer`` and therefore ``bench.py`` start from)."""
    pcfg = plbert_amd.AlbertConfig(**cfg_kwargs)
    gen = plbert_amd.deterministic_state_dict if init == "deterministic" else plbert_amd.reference_init_state_dict
    sd = gen(pcfg, num_phonemes, num_tokens, seed=seed)
    labels, masked, lengths, idxs = batch
    out = dict(labels=labels, masked=masked, lengths=np.array(lengths), index=obj_array(idxs),
               seed=np.array(seed), init=np.array(init), lr=np.array(lr),
               num_phonemes=np.array(num_phonemes), num_tokens=np.array(num_tokens),
               cfg_keys=np.array(list(cfg_kwargs.keys())), cfg_vals=np.array(list(cfg_kwargs.values())))

    m = build_reference(cfg_kwargs, num_phonemes, num_tokens, sd)
    m.train()
    tl, tm = torch.from_numpy(labels), torch.from_numpy(masked)
    text_mask = ref_train.length_to_mask(torch.Tensor(lengths))
    am = (~text_mask).int()
    with torch.no_grad():
        pred = m(tm, attention_mask=am)
        hidden = m.encoder(tm, attention_mask=am).last_hidden_state
    if num_tokens:
        pred, tok_pred = pred
        out["token_logits"] = tok_pred.numpy() if full else tok_pred.numpy()[:, :4, :64]
    # sdpa delta, for the record
    m_sdpa = build_reference(cfg_kwargs, num_phonemes, num_tokens, sd, attn="sdpa")
    with torch.no_grad():
        p2 = m_sdpa(tm, attention_mask=am)
        p2 = p2[0] if num_tokens else p2
    valid = am.bool().numpy()
    out["sdpa_max_abs_delta_valid_rows"] = np.array(np.abs(p2.numpy() - pred.numpy())[valid].max())

    if not num_tokens:
        # the training step of the reference: process_batch -> backward -> AdamW (train.py:350-357)
        crit = torch.nn.CrossEntropyLoss()
        opt = torch.optim.AdamW(m.parameters(), lr=lr)
        losses = []
        for step in range(n_steps):
            loss = ref_train.process_batch(m, (tl, tm, lengths, idxs), crit, _Acc())
            opt.zero_grad()
            loss.backward()
            if step == 0:
                grads = {k: (p.grad.detach().numpy().copy() if p.grad is not None else None)
                         for k, p in m.named_parameters()}
            opt.step()
            losses.append(float(loss.item()))
        out["losses"] = np.array(losses, dtype=np.float64)
        out["loss"] = np.array(losses[0], dtype=np.float64)
        names = [k for k, g in grads.items() if g is not None]
        out["grad_names"] = np.array(names)
        out["grad_none_names"] = np.array([k for k, g in grads.items() if g is None])
        out["grad_l2"] = np.array([np.sqrt((grads[k].astype(np.float64) ** 2).sum()) for k in names])
        final = {k: v.detach().numpy() for k, v in m.state_dict().items()}
        out["final_param_l2"] = np.array([np.sqrt((final[k].astype(np.float64) ** 2).sum()) for k in sd])
        out["param_names"] = np.array(list(sd.keys()))
        if full:
            for k in names:
                out["grad/" + k] = grads[k]
            for k in sd:
                out["final/" + k] = final[k]
        else:
            rs = np.random.RandomState(99)
            for k in names:
                flat = grads[k].reshape(-1)
                probe = rs.randint(0, flat.size, size=min(8, flat.size))
                out["gprobe_idx/" + k] = probe
                out["gprobe_val/" + k] = flat[probe]

    if full:
        out["logits"] = pred.numpy()
        out["hidden"] = hidden.numpy()
    else:
        B, S = labels.shape
        rs = np.random.RandomState(5)
        pb = rs.randint(0, B, size=16)
        ps = np.array([rs.randint(0, lengths[b]) for b in pb])
        out["probe_b"], out["probe_s"] = pb, ps
        out["probe_logits"] = pred.numpy()[pb, ps, :]
        out["probe_hidden"] = hidden.numpy()[pb, ps, :]
        out["logit_row_sums"] = pred.numpy().sum(-1)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out, allow_pickle=True)
    print(tag, "written; loss", out.get("loss"), "sdpa delta", out["sdpa_max_abs_delta_valid_rows"])


def capture_dualloss(tag, cfg_kwargs, num_phonemes, num_tokens, batch, seed, n_steps, lr=1e-3, full=True, init="deterministic",
                     token_ids=None):
    """Dual-head training on the reference's MultiTaskModel: loss = calculate_phoneme_loss (train.py:107-131)
    + the token loss of oracle.albert_np.token_loss (per-sample mean CE over [:length], mean over samples —
    upstream PL-BERT's loss_vocab; the reference itself has no token loss). torch autograd + AdamW."""
    pcfg = plbert_amd.AlbertConfig(**cfg_kwargs)
    gen = plbert_amd.deterministic_state_dict if init == "deterministic" else plbert_amd.reference_init_state_dict
    sd = gen(pcfg, num_phonemes, num_tokens, seed=seed)
    labels, masked, lengths, idxs = batch
    if token_ids is None:
        rs = np.random.RandomState(seed + 1000)
        token_ids = np.zeros_like(labels)
        for b, L in enumerate(lengths):
            token_ids[b, :L] = rs.randint(0, num_tokens, size=L)
    out = dict(labels=labels, masked=masked, lengths=np.array(lengths), index=obj_array(idxs), token_ids=token_ids,
               seed=np.array(seed), num_phonemes=np.array(num_phonemes), num_tokens=np.array(num_tokens),
               cfg_keys=np.array(list(cfg_kwargs.keys())), cfg_vals=np.array(list(cfg_kwargs.values())))
    if not full:   # (the small fixtures predate these two entries and stay byte-identical on regeneration)
        out["init"], out["lr"] = np.array(init), np.array(lr)
    m = build_reference(cfg_kwargs, num_phonemes, num_tokens, sd)
    m.train()
    tl, tm, tt = torch.from_numpy(labels), torch.from_numpy(masked), torch.from_numpy(token_ids)
    am = (~ref_train.length_to_mask(torch.Tensor(lengths))).int()
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.AdamW(m.parameters(), lr=lr)
    losses, parts = [], []
    for step in range(n_steps):
        ph, tk = m(tm, attention_mask=am)
        lp = ref_train.calculate_phoneme_loss(ph, tl, lengths, idxs, crit)
        lt = 0
        for pred_b, tgt_b, L in zip(tk, tt, lengths):
            lt = lt + crit(pred_b[:L], tgt_b[:L])
        lt = lt / tk.size(0)
        loss = lp + lt
        opt.zero_grad()
        loss.backward()
        if step == 0:
            grads = {k: (p.grad.detach().numpy().copy() if p.grad is not None else None) for k, p in m.named_parameters()}
        opt.step()
        losses.append(float(loss.item()))
        parts.append([float(lp.item()), float(lt.item())])
    out["losses"] = np.array(losses, dtype=np.float64)
    out["loss_parts"] = np.array(parts, dtype=np.float64)
    names = [k for k, g in grads.items() if g is not None]
    out["grad_names"] = np.array(names)
    out["grad_none_names"] = np.array([k for k, g in grads.items() if g is None])
    final = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    if full:
        for k in names:
            out["grad/" + k] = grads[k]
        out["param_names"] = np.array(list(sd.keys()))
        for k in sd:
            out["final/" + k] = final[k]
    else:   # probes only (full-size captures): norms, 8 elements per gradient tensor, 16 rows of both heads' first logits
        out["param_names"] = np.array(list(sd.keys()))
        out["grad_l2"] = np.array([np.sqrt((grads[k].astype(np.float64) ** 2).sum()) for k in names])
        out["final_param_l2"] = np.array([np.sqrt((final[k].astype(np.float64) ** 2).sum()) for k in sd])
        rp = np.random.RandomState(99)
        for k in names:
            flat = grads[k].reshape(-1)
            probe = rp.randint(0, flat.size, size=min(8, flat.size))
            out["gprobe_idx/" + k] = probe
            out["gprobe_val/" + k] = flat[probe]
        m0 = build_reference(cfg_kwargs, num_phonemes, num_tokens, sd)
        with torch.no_grad():
            ph0, tk0 = m0(tm, attention_mask=am)
        B = labels.shape[0]
        rq = np.random.RandomState(5)
        pb = rq.randint(0, B, size=16)
        ps = np.array([rq.randint(0, lengths[b]) for b in pb])
        out["probe_b"], out["probe_s"] = pb, ps
        out["probe_logits"] = ph0.numpy()[pb, ps, :]
        out["probe_token_logits"] = tk0.numpy()[pb, ps, :]
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out, allow_pickle=True)
    print(tag, "written; losses", losses, "parts", parts[0])


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else ""   # "dualloss" / "large" / "fullsize_a" / "fullsize_d" / "fullsize_ragged" / "fullsize_dual" / "fullsize_b96": only those
    real = dict(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                max_position_embeddings=512, num_hidden_layers=12)
    # (4) BASELINE configs[3]'s architecture (hidden 1024, 24 shared layers, 16 heads, FFN 4096; SURVEY.md section 8:
    # ALBERT-large's heads / FFN width) at 4 x 256 = 1024 tokens with one ragged row: the token count at which the
    # engine runs its fused LayerNorm epilogues with FOUR column tiles per row block (H = 1024); probes only
    large = dict(vocab_size=188, hidden_size=1024, num_attention_heads=16, intermediate_size=4096,
                 max_position_embeddings=512, num_hidden_layers=24)
    # (5) the sizes BASELINE.json configs[1] / configs[3] state, on EXACTLY bench.py's rank-0 inputs: reference
    # initialisation seed 0 (PLBertTrainer's default), synthetic_batch(B, 512, seed=1234), AdamW lr 7e-5 — probes only
    # (loss, 16 logit rows, per-parameter gradient norms + probes, AdamW loss trajectory). The reference's step on these
    # batches is 30-70 s of CPU; they are captured on request only ("fullsize_a" / "fullsize_d"), not in the default run.
    if only == "fullsize_a":
        capture_model("real_s512_b32", real, 188, 0, plbert_amd.synthetic_batch(32, 512, seed=1234), seed=0,
                      full=False, n_steps=5, init="reference")
        return
    if only == "fullsize_d":
        capture_model("real_h1024_s512_b16", large, 188, 0, plbert_amd.synthetic_batch(16, 512, seed=1234), seed=0,
                      full=False, n_steps=2, init="reference")
        return
    if only == "fullsize_ragged":
        # the same size with RAGGED samples (N2, dataloader.py:276-297): 32 lengths between 64 and 512, longest first as the
        # collater sorts them — attention tile skipping, padded rows and the pruned last application at 16,384 rows
        rs = np.random.RandomState(77)
        lengths = sorted([512] + rs.randint(64, 513, size=31).tolist(), reverse=True)
        capture_model("real_s512_b32_ragged", real, 188, 0, ragged_batch(32, 512, lengths, seed=31), seed=0,
                      full=False, n_steps=2, init="reference")
        return
    if only == "fullsize_b96":
        # configs/config.yml's own batch_size (96 x 512 = 49,152 rows: 384 row blocks, 1,152 attention items, 589,824 stacked
        # rows in the weight-gradient GEMMs), reference initialisation, 2 AdamW steps; probes only (~90 s of CPU per step)
        capture_model("real_s512_b96", real, 188, 0, plbert_amd.synthetic_batch(96, 512, seed=1234), seed=0,
                      full=False, n_steps=2, init="reference")
        return
    if only == "fullsize_dual":
        # BASELINE configs[1]'s "dual-head loss" at its stated size: MultiTaskModel 768/12 with a 5,000-token head (not a
        # multiple of the 256-column tile of the fused GEMM + cross-entropy passes), bench.py's batch and initialisation,
        # torch autograd over the reference model for the gradients, AdamW lr 7e-5, 2 steps; probes only
        # (token targets as bench.py --num-tokens draws them on rank 0, so that its first steps are comparable)
        capture_dualloss("real_s512_b32_dualloss", real, 188, 5000, plbert_amd.synthetic_batch(32, 512, seed=1234), seed=0,
                         n_steps=2, lr=7e-5, full=False, init="reference",
                         token_ids=np.random.RandomState(4321).randint(0, 5000, size=(32, 512)).astype(np.int64))
        return
    if only in ("", "large"):
        capture_model("real_h1024_s256_b4", large, 188, 0, ragged_batch(4, 256, [256, 256, 256, 201], seed=8), seed=24,
                      full=False, n_steps=2)
        if only:
            return
    tiny = dict(vocab_size=188, embedding_size=16, hidden_size=64, num_attention_heads=4,
                intermediate_size=128, num_hidden_layers=2, max_position_embeddings=512)
    small = dict(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                 intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    if only in ("", "dualloss"):
        # (2c) dual-head training (phoneme + token loss), tiny and small
        capture_dualloss("tiny_h64_dualloss", tiny, 188, 40, ragged_batch(2, 16, [16, 11], seed=3), seed=21, n_steps=3)
        capture_dualloss("small_h128_dualloss", small, 188, 256, ragged_batch(3, 40, [40, 33, 7], seed=4), seed=22,
                         n_steps=3)
        if only:
            return
    gen_masking()
    # (2a) tiny model of SURVEY.md §4: everything stored in full (oracle pinning)
    tiny = dict(vocab_size=188, embedding_size=16, hidden_size=64, num_attention_heads=4,
                intermediate_size=128, num_hidden_layers=2, max_position_embeddings=512)
    capture_model("tiny_h64", tiny, 188, 0, ragged_batch(2, 16, [16, 11], seed=3), seed=21, full=True, n_steps=3,
                  lr=1e-3)
    capture_model("tiny_h64_multitask", tiny, 188, 40, ragged_batch(2, 16, [16, 11], seed=3), seed=21, full=True,
                  n_steps=0)
    # (2b) smallest shape the HIP kernels accept (head_dim 64): full tensors, ragged lengths
    small = dict(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                 intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    capture_model("small_h128", small, 188, 0, ragged_batch(3, 40, [40, 33, 7], seed=4), seed=22, full=True,
                  n_steps=3, lr=1e-3)
    capture_model("small_h128_multitask", small, 188, 96, ragged_batch(3, 40, [40, 33, 7], seed=4), seed=22,
                  full=True, n_steps=0)
    # (3) the real 768/12 model (configs/config.yml:32-39); weights regenerated on both sides, not stored
    lab, msk, lens, idx = plbert_amd.synthetic_batch(8, 128, seed=1234)
    capture_model("real_s128_b8", real, 188, 0, (lab, msk, lens, idx), seed=23, full=False, n_steps=5)
    capture_model("real_s512_b2_ragged", real, 188, 0, ragged_batch(2, 512, [512, 300], seed=6), seed=23,
                  full=False, n_steps=2)


if __name__ == "__main__":
    main()
