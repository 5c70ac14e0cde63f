"""Masking / index path: bit-exact against vectors captured from the reference's dataloader.py."""

import numpy as np
import torch

from conftest import load_golden
import plbert_amd
from plbert_amd import data as pdata

PARAMS = dict(word_separator=87, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1)


def _docs(g):
    return [d.split("\x1f") for d in g["docs"]]


def test_symbol_table():
    g = load_golden("masking")
    assert [ord(c) for c in plbert_amd.symbols] == g["symbols_codepoints"].tolist()
    ci = plbert_amd.CharacterIndexer()
    assert ci("P M¤U") == [0, 186, 185, 187, 187]


def _eq_lists(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64))


def test_getitem_stream_and_collate_3tuple():
    g = load_golden("masking")
    data = [{"phonemes": d} for d in _docs(g)]
    for tag, msl in (("msl512", 512), ("msl32", 32)):
        pdata.seed_reference_streams(1)
        ds = plbert_amd.MaskedPhonemeDataset(data, max_seq_length=msl, use_token_ids=False, **PARAMS)
        items = [ds[int(i)] for i in g[f"{tag}_order"]]
        _eq_lists([i[0].numpy() for i in items], g[f"{tag}_labels"])
        _eq_lists([i[1].numpy() for i in items], g[f"{tag}_masked"])
        _eq_lists([i[2] for i in items], g[f"{tag}_index"])
        assert all(i[0].dtype == torch.int64 and i[1].dtype == torch.int64 for i in items)
        lab, msk, lens, idx = plbert_amd.PhonemeOnlyCollater()(items[:8])
        assert np.array_equal(lab.numpy(), g[f"{tag}_c3_labels"])
        assert np.array_equal(msk.numpy(), g[f"{tag}_c3_masked"])
        assert lens == g[f"{tag}_c3_lengths"].tolist()
        _eq_lists(idx, g[f"{tag}_c3_index"])
        tm = plbert_amd.length_to_mask(torch.Tensor(lens))
        assert tm.dtype == torch.bool and np.array_equal(tm.numpy(), g[f"{tag}_c3_text_mask"])


def test_collate_4tuple_with_token_ids():
    g = load_golden("masking")
    data = [{"phonemes": d, "token_ids": t.tolist()} for d, t in zip(_docs(g), g["token_ids"])]
    for tag, msl in (("msl512", 512), ("msl32", 32)):
        pdata.seed_reference_streams(1)
        ds = plbert_amd.MaskedPhonemeDataset(data, max_seq_length=msl, use_token_ids=True, **PARAMS)
        items = [ds[int(i)] for i in g[f"{tag}_order"][:8]]
        tok, lab, msk, lens, idx = plbert_amd.Collater()(items)
        assert np.array_equal(tok.numpy(), g[f"{tag}_c4_tokens"])
        assert np.array_equal(lab.numpy(), g[f"{tag}_c4_labels"])
        assert np.array_equal(msk.numpy(), g[f"{tag}_c4_masked"])
        assert lens == g[f"{tag}_c4_lengths"].tolist()
        _eq_lists(idx, g[f"{tag}_c4_index"])


def test_survey_known_answer():
    """mask id 185, separator 186 never indexed, crop re-bases indices."""
    g = load_golden("masking")
    toy = [{"phonemes": ["aa", "bb", "cc", "dd", "ee", "ff", "gg", "hh"]}, {"phonemes": ["ab", "cd", "ef"]}]
    pdata.seed_reference_streams(1)
    ds = plbert_amd.MaskedPhonemeDataset(toy, max_seq_length=16, use_token_ids=False, **PARAMS)
    its = [ds[int(i)] for i in g["ka_calls"]]
    _eq_lists([i[0].numpy() for i in its], g["ka_labels"])
    _eq_lists([i[1].numpy() for i in its], g["ka_masked"])
    _eq_lists([i[2] for i in its], g["ka_index"])
    for lab, msk, idx in its:
        assert all(lab[i] != 186 for i in idx)
        changed = (lab != msk).nonzero().flatten().tolist()
        assert set(changed) <= set(idx)


def test_train_val_split_order():
    g = load_golden("masking")
    pdata.seed_reference_streams(1)
    big = [{"phonemes": ["ab", "cd"]} for _ in range(50)]
    tl, vl = plbert_amd.build_dataloader(big, batch_size=4, device="cpu",
                                         dataset_config=dict(max_seq_length=512, **PARAMS), use_token_ids=False)
    assert list(tl.dataset.indices) == g["split_train_indices"].tolist()
    assert list(vl.dataset.indices) == g["split_val_indices"].tolist()


def test_edge_cases():
    pdata.seed_reference_streams(3)
    ds = plbert_amd.MaskedPhonemeDataset([{"phonemes": []}, {"phonemes": ["a"]}], max_seq_length=4,
                                         use_token_ids=False, **PARAMS)
    lab, msk, idx = ds[0]
    assert lab.numel() == 0 and msk.numel() == 0 and idx == []
    lab, msk, idx = ds[1]
    assert lab.tolist() == [159, 186]
    off, flat = plbert_amd.masked_indices_to_csr([[1, 2], [], [0]])
    assert off.tolist() == [0, 2, 2, 3] and flat.tolist() == [1, 2, 0]
    l, m, lens, ix = plbert_amd.synthetic_batch(4, 64, seed=5)
    assert l.shape == (4, 64) and all(len(i) > 0 for i in ix) and lens == [64] * 4
    assert ((l != m).sum(1) > 0).all() and (m[l == 186] == 186).all()


def test_multi_worker_streams_are_independent_and_reproducible():
    """N3: with num_workers > 0 every worker re-seeds BOTH masking streams from its torch worker seed (data.seed_worker).
    Without that, forked workers share the parent's NumPy state and draw identical masks (the reference never runs
    workers: train.py:253). Same (torch seed, num_workers) -> same batches; different workers -> different draws."""
    docs = [{"phonemes": ["abcde", "fghij", "klmno", "pqrst"] * 30} for _ in range(64)]   # identical documents
    cfg = dict(max_seq_length=128, **PARAMS)

    def run(seed):
        torch.manual_seed(seed)
        pdata.seed_reference_streams(1)
        tl, _ = plbert_amd.build_dataloader(docs, batch_size=4, device="cpu", dataset_config=cfg, use_token_ids=False,
                                            num_workers=2)
        out = []
        for i, (lab, msk, lens, idx) in enumerate(tl):
            out.append((msk.numpy().copy(), [list(x) for x in idx]))
            if i == 3:
                break
        del tl
        return out

    a, b, c = run(7), run(7), run(8)
    for (m1, i1), (m2, i2) in zip(a, b):
        assert np.array_equal(m1, m2) and i1 == i2            # reproducible
    assert any(not np.array_equal(m1, m2) for (m1, _), (m2, _) in zip(a, c))   # the torch seed drives the masks
    # batches 0 and 1 come from workers 0 and 1: identical documents, so equal masks would mean a shared stream
    assert not np.array_equal(a[0][0], a[1][0])
    # within a batch, samples are not copies of each other either
    assert len({tuple(r) for r in a[0][0].tolist()}) > 1


def test_decisions_loader_collates_for_the_device_path():
    g = load_golden("masking")
    data = [{"phonemes": d} for d in _docs(g)] * 4
    pdata.seed_reference_streams(1)
    tl, vl = plbert_amd.build_dataloader(data, batch_size=4, device="cpu", dataset_config=dict(max_seq_length=64, **PARAMS),
                                         use_token_ids=False, decisions=True)
    c = next(iter(tl))
    assert set(c) >= {"ids", "repl", "sample_off", "word_off", "word_begin", "word_len", "action", "crop_start", "lengths"}
    assert c["B"] == 4 and c["S"] == max(c["lengths"]) <= 64 and c["lengths"] == sorted(c["lengths"], reverse=True)
    assert c["sample_off"][-1] == len(c["ids"]) == len(c["repl"]) and c["word_off"][-1] == len(c["action"])
