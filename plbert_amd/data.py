"""Host side of the masking / index path (SURVEY.md §8(a) rows A1–A3), bit-exact with the reference.

Follows ``dataloader.py:19-142`` (MaskedPhonemeDataset), ``:200-223`` (Collater), ``:276-297``
(PhonemeOnlyCollater), ``:225-274`` (build_dataloader) and ``train.py:34-44`` (length_to_mask).

Bit-exactness depends on drawing from the SAME global generators in the SAME order as the
reference: one ``np.random.rand()`` per word; for a selected word one ``np.random.choice`` over
(mask, replace, keep); for a replaced word ``len(word)`` draws of the stdlib ``random`` stream; one
``np.random.randint`` per cropped sample.  Like ``dataloader.py:16-17`` this module seeds both
streams with 1 when imported.  Everything downstream of the decisions is integer work on id arrays.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from .symbols import CharacterIndexer, MASK_ID, SEPARATOR_ID


def seed_reference_streams(seed=1):
    np.random.seed(seed)
    random.seed(seed)


seed_reference_streams(1)  # dataloader.py:16-17

_ACTIONS = ("mask", "replace", "no_change")


class MaskedPhonemeDataset(torch.utils.data.Dataset):
    """Word-level mask / replace / keep over phoneme words (dataloader.py:19-142).

    A row of ``dataset`` is ``{'phonemes': list[str]}`` (+ ``'token_ids'`` when ``use_token_ids``).
    ``__getitem__`` returns ``(labels, masked, masked_index)`` or, with token ids,
    ``(token_ids, labels, masked, masked_index)``; tensors are int64, ``masked_index`` a list.
    """

    def __init__(self, dataset, word_pred_prob, phoneme_mask_prob, replace_prob, word_separator,
                 max_seq_length, use_token_ids):
        self.data = dataset
        self.max_seq_length = max_seq_length
        self.word_pred_prob = word_pred_prob
        self.phoneme_mask_prob = phoneme_mask_prob
        self.replace_prob = replace_prob
        self.char_indexer = CharacterIndexer()
        self.word_separator = word_separator
        self.use_token_ids = use_token_ids
        # same float expression as dataloader.py:88 so the cumulative table is identical
        self._probs = [phoneme_mask_prob, replace_prob, 1 - (phoneme_mask_prob + replace_prob)]

    def __len__(self):
        return len(self.data)

    def _decide(self, words, pool_ids):
        """The reference's random draws for one document, in the reference's order, as DECISIONS:
        (ids, word_begin, word_len, action, repl) with ``ids`` the uncropped labels (a separator after each
        word), ``action[w]`` 0 keep / 1 mask / 2 replace, ``repl`` the replacement ids at the positions of
        replaced words (0 elsewhere). Applying them is integer work — here (``_apply``) or on the device
        (plb_apply_mask)."""
        index_word = self.char_indexer
        ids_all, begin, length, action, repl = [], [], [], [], []
        n_pool = len(pool_ids)
        for w in words:
            ids = index_word(w)
            n = len(ids)
            begin.append(len(ids_all))
            length.append(n)
            act = 0
            r = [0] * n
            if np.random.rand() < self.word_pred_prob:                       # dataloader.py:85
                a = _ACTIONS[int(np.random.choice(3, p=self._probs))]       # dataloader.py:89
                if a == "replace":
                    # random.choices(pool, k=n) draws floor(random() * len(pool)) n times (dataloader.py:94)
                    picks = random.choices(range(n_pool), k=n) if n else []
                    r = [pool_ids[i] for i in picks]
                    act = 2
                elif a == "mask":
                    act = 1
            action.append(act)
            ids_all.extend(ids)
            ids_all.append(SEPARATOR_ID)
            repl.extend(r)
            repl.append(0)
        return ids_all, begin, length, action, repl

    @staticmethod
    def _apply(ids_all, begin, length, action, repl):
        """Decisions -> (labels, masked, index) id lists (dataloader.py:66-104; the separator is never indexed)."""
        masked = list(ids_all)
        index = []
        for b, n, a in zip(begin, length, action):
            if a == 1:
                masked[b:b + n] = [MASK_ID] * n
            elif a == 2:
                masked[b:b + n] = repl[b:b + n]
            if a:
                index.extend(range(b, b + n))
        return list(ids_all), masked, index

    def decisions(self, idx):
        """Everything ``__getitem__`` draws for row ``idx`` — consuming the global streams exactly as it does — without
        applying it: a dict of int arrays for ``device_apply_mask`` (plb_apply_mask does the integer work on the GPU).
        ``length`` is the sample's collated length min(n, max_seq_length)."""
        row = self.data[idx]
        words = row["phonemes"]
        pool_ids = self.char_indexer("".join(words))
        ids_all, begin, length, action, repl = self._decide(words, pool_ids)
        n = len(ids_all)
        start = 0
        if n > self.max_seq_length:                                         # dataloader.py:110-112
            start = int(np.random.randint(0, n - self.max_seq_length))
        tok = row["token_ids"] if self.use_token_ids else None
        return {"ids": np.asarray(ids_all, dtype=np.int64), "word_begin": np.asarray(begin, dtype=np.int32),
                "word_len": np.asarray(length, dtype=np.int32), "action": np.asarray(action, dtype=np.int8),
                "repl": np.asarray(repl, dtype=np.int64), "crop_start": start,
                "length": min(n, self.max_seq_length),
                "word_token": None if tok is None else np.asarray(tok, dtype=np.int64)[: len(words)]}

    def __getitem__(self, idx):
        row = self.data[idx]
        words = row["phonemes"]
        pool_ids = self.char_indexer("".join(words))  # replacement pool: the document itself
        tok = row["token_ids"] if self.use_token_ids else [self.word_separator] * len(words)

        labels, masked, index = self._apply(*self._decide(words, pool_ids))
        token_ids = []
        for w, t in zip(words, tok):
            token_ids.extend([t] * len(w))
            token_ids.append(self.word_separator)

        n = len(masked)
        if n > self.max_seq_length:                                         # dataloader.py:110-126
            start = int(np.random.randint(0, n - self.max_seq_length))
            end = start + self.max_seq_length
            masked, token_ids, labels = masked[start:end], token_ids[start:end], labels[start:end]
            index = [i - start for i in index if start <= i < end]

        masked_t = torch.LongTensor(masked)
        labels_t = torch.LongTensor(labels)
        token_t = torch.LongTensor(token_ids)
        assert len(masked_t) == len(token_t) == len(labels_t)
        if self.use_token_ids:
            return token_t, labels_t, masked_t, index
        return labels_t, masked_t, index


def collate_decisions(records):
    """The collaters' ordering (dataloader.py:204,280: sort by length, descending, stable) applied to decision
    records, packed into the flat arrays plb_apply_mask takes. Returns a dict of numpy arrays + B, S."""
    recs = sorted(records, key=lambda r: r["length"], reverse=True)
    B = len(recs)
    S = int(recs[0]["length"]) if B else 0
    sample_off = np.zeros(B + 1, dtype=np.int32)
    word_off = np.zeros(B + 1, dtype=np.int32)
    for b, r in enumerate(recs):
        sample_off[b + 1] = sample_off[b] + len(r["ids"])
        word_off[b + 1] = word_off[b] + len(r["word_begin"])
    cat = lambda k, dt: (np.concatenate([r[k] for r in recs]).astype(dt) if B else np.zeros(0, dt))
    with_tok = B > 0 and recs[0]["word_token"] is not None
    return {"B": B, "S": S, "ids": cat("ids", np.int64), "repl": cat("repl", np.int64), "sample_off": sample_off,
            "word_off": word_off, "word_begin": cat("word_begin", np.int32), "word_len": cat("word_len", np.int32),
            "action": cat("action", np.int8), "crop_start": np.asarray([r["crop_start"] for r in recs], dtype=np.int32),
            "lengths": [int(r["length"]) for r in recs],
            "word_token": cat("word_token", np.int64) if with_tok else None}


def _pad_stack(rows, width):
    out = torch.zeros((len(rows), width), dtype=torch.long)  # pad id 0 = 'P'
    for i, r in enumerate(rows):
        out[i, : r.shape[0]] = r
    return out


class PhonemeOnlyCollater:
    """Length-descending sort + zero pad to the batch maximum (dataloader.py:276-297)."""

    def __call__(self, batch):
        batch = sorted(batch, key=lambda x: x[0].shape[0], reverse=True)
        width = batch[0][0].shape[0]
        labels = _pad_stack([b[0] for b in batch], width)
        masked = _pad_stack([b[1] for b in batch], width)
        lengths = [int(b[1].shape[0]) for b in batch]
        indices = [b[2] for b in batch]
        return labels, masked, lengths, indices


class Collater:
    """4-tuple form with token ids (dataloader.py:200-223)."""

    def __call__(self, batch):
        batch = sorted(batch, key=lambda x: x[0].shape[0], reverse=True)
        width = batch[0][0].shape[0]
        tokens = _pad_stack([b[0] for b in batch], width)
        labels = _pad_stack([b[1] for b in batch], width)
        masked = _pad_stack([b[2] for b in batch], width)
        lengths = [int(b[2].shape[0]) for b in batch]
        indices = [b[3] for b in batch]
        return tokens, labels, masked, lengths, indices


def seed_worker(worker_id):
    """``worker_init_fn`` of a multi-worker DataLoader over MaskedPhonemeDataset. The masking draws from NumPy's and
    Python's GLOBAL generators (dataloader.py:85-94); forked workers all inherit the parent's NumPy state, so without
    this every worker would draw the SAME masks for its samples. torch gives each worker the seed
    ``base_seed + worker_id`` (and seeds ``random`` and ``torch`` with it, not NumPy): both streams of the masking are
    re-seeded from it here, so a run is reproducible given (torch seed, num_workers) and workers are independent.
    With ``num_workers=0`` nothing is re-seeded: the stream is the reference's, bit for bit (tests/golden/masking.npz)."""
    info = torch.utils.data.get_worker_info()
    seed = (info.seed if info is not None else torch.initial_seed()) % (2 ** 32)
    np.random.seed(seed)
    random.seed(seed)


class DecisionsDataset(torch.utils.data.Dataset):
    """View of a MaskedPhonemeDataset whose items are ``decisions(i)`` records — the random draws only — for the
    device-side application of the masking (plb_apply_mask): the host workers do the RNG and the char indexing, the GPU
    the cropping, substitution, index lists and padding."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset.decisions(idx)


def build_dataloader(df, batch_size, device, dataset_config, use_token_ids, num_workers=0, decisions=False, shard=None,
                     **kwargs):
    """95/5 train/val split by ``random.shuffle`` then two DataLoaders (dataloader.py:225-274).
    ``shard=(rank, world)`` (not in the reference, whose every rank loads — and masks — the WHOLE global batch and keeps a
    slice, accelerate split_batches=True): this rank's loaders hold every world-th sample of the split and batches of
    ``batch_size // world``, so the host work per rank is 1 / world of the reference's; the global batch is still
    ``batch_size`` distinct samples per step, but which samples meet in it differs from the reference's order.
    ``num_workers > 0`` (the reference runs 0, train.py:253 — the single-threaded Python masking is its input
    bottleneck): worker processes with independent, reproducible masking streams (``seed_worker``).
    ``decisions=True``: batches are ``collate_decisions`` dicts for ``plbert_amd.pipeline.DeviceFeeder`` /
    ``device_apply_mask`` instead of collated tensors."""
    from torch.utils.data import DataLoader, Subset

    dataset = MaskedPhonemeDataset(df, use_token_ids=use_token_ids, **dataset_config)
    total = len(dataset)
    val_size = min(int(total * 0.05), 10000)
    order = list(range(total))
    random.shuffle(order)
    train_idx, val_idx = order[: total - val_size], order[total - val_size:]
    if shard is not None:
        rank, world = shard
        if batch_size % world:
            raise ValueError(f"batch_size {batch_size} is not divisible by the world size {world}")
        train_idx, val_idx, batch_size = train_idx[rank::world], val_idx[rank::world], batch_size // world
    train_set = Subset(dataset, train_idx)
    val_set = Subset(dataset, val_idx)
    collate = Collater() if use_token_ids else PhonemeOnlyCollater()
    pin = device != "cpu"
    if decisions:
        train_set = Subset(DecisionsDataset(dataset), train_idx)
        val_set = Subset(DecisionsDataset(dataset), val_idx)
        collate, pin = collate_decisions, False
    if num_workers > 0:
        kwargs.setdefault("worker_init_fn", seed_worker)
        kwargs.setdefault("persistent_workers", True)
    train_loader = DataLoader(train_set, batch_size=batch_size, shuffle=True, drop_last=True,
                              collate_fn=collate, pin_memory=pin, num_workers=num_workers, **kwargs)
    val_loader = DataLoader(val_set, batch_size=batch_size, shuffle=False, drop_last=False,
                            collate_fn=collate, pin_memory=pin, num_workers=num_workers, **kwargs)
    if shard is not None:
        train_loader.plb_per_rank = val_loader.plb_per_rank = True   # (run._source: nothing left to slice)
    return train_loader, val_loader


def length_to_mask(lengths):
    """train.py:34-44 — bool [B, max_len], True on PAD: (position + 1) > length."""
    lengths = torch.as_tensor(lengths)
    width = int(lengths.max().item())
    pos = torch.arange(width, device=lengths.device).expand(lengths.size(0), width)
    return (pos + 1) > lengths.unsqueeze(1)


def masked_indices_to_csr(masked_indices):
    """list[list[int]] -> (offsets int32 [B+1], flat int32 [n]) for the device loss kernels."""
    offsets = np.zeros(len(masked_indices) + 1, dtype=np.int32)
    for b, idx in enumerate(masked_indices):
        offsets[b + 1] = offsets[b] + len(idx)
    flat = np.fromiter((i for idx in masked_indices for i in idx), dtype=np.int32, count=int(offsets[-1]))
    return offsets, flat


def synthetic_batch(batch_size, seq_len, seed, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1):
    """Fixed-length synthetic batch of SURVEY.md §8(d): ids uniform over 1..184, words of length
    3..7 closed by the separator 186, word-level masking with the reference probabilities, every
    sample guaranteed at least one masked index.  Own RandomState: global streams untouched."""
    rs = np.random.RandomState(seed)
    labels = np.zeros((batch_size, seq_len), dtype=np.int64)
    masked = np.zeros((batch_size, seq_len), dtype=np.int64)
    indices = []
    cdf = np.cumsum([phoneme_mask_prob, replace_prob, 1 - (phoneme_mask_prob + replace_prob)])
    for b in range(batch_size):
        lab = rs.randint(1, 185, size=seq_len).astype(np.int64)
        bounds, pos = [], 0
        while pos < seq_len:
            n = int(rs.randint(3, 8))
            end = min(pos + n, seq_len)
            bounds.append((pos, end))
            if end < seq_len:
                lab[end] = SEPARATOR_ID
            pos = end + 1
        msk = lab.copy()
        idx = []
        for (s, e) in bounds:
            if rs.rand() < word_pred_prob:
                a = int(np.searchsorted(cdf, rs.rand(), side="right"))
                if a == 0:
                    msk[s:e] = MASK_ID
                elif a == 1:
                    msk[s:e] = lab[rs.randint(0, seq_len, size=e - s)]
                if a != 2:
                    idx.extend(range(s, e))
        if not idx:
            s, e = bounds[0]
            msk[s:e] = MASK_ID
            idx = list(range(s, e))
        labels[b], masked[b] = lab, msk
        indices.append(idx)
    return labels, masked, [seq_len] * batch_size, indices
