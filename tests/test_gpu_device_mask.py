"""Device-side masking fast mode (-m gpu): distribution-matched to dataloader.py:83-108, structural
invariants exact, reproducible from (seed, step)."""
import numpy as np
import pytest
import torch

import plbert_amd

pytestmark = pytest.mark.gpu


def _labels(B, S, seed):
    rs = np.random.RandomState(seed)
    lab = np.zeros((B, S), np.int64)
    lengths = []
    for b in range(B):
        L = S if b % 3 else int(rs.randint(S // 2, S + 1))
        pos = 0
        while pos < L:
            n = int(rs.randint(1, 8))
            end = min(pos + n, L)
            lab[b, pos:end] = rs.randint(1, 185, size=end - pos)
            if end < L:
                lab[b, end] = 186
            pos = end + 1
        lengths.append(L)
    return lab, lengths


def _words(row, L):
    out, start = [], None
    for i in range(L):
        if row[i] == 186:
            if start is not None:
                out.append((start, i)); start = None
        elif start is None:
            start = i
    if start is not None:
        out.append((start, L))
    return out


def test_structure_and_distribution():
    B, S = 96, 512
    lab, lengths = _labels(B, S, 5)
    sb = plbert_amd.device_mask_batch(lab, lengths, seed=1234, step=7)
    torch.cuda.synchronize()
    msk = sb.masked.cpu().numpy()
    off = sb.offsets.cpu().numpy()
    flat = sb.flat.cpu().numpy()
    assert off[0] == 0 and off[-1] == sb.n_masked == len(flat) and (np.diff(off) >= 0).all()
    n_words = n_sel = n_mask = n_repl = 0
    for b in range(B):
        L = lengths[b]
        idx = flat[off[b]:off[b + 1]]
        assert (np.diff(idx) > 0).all() and (idx < L).all()               # ascending, inside the length
        assert (lab[b, idx] != 186).all()                                 # separators never indexed
        keep = np.ones(S, bool); keep[idx] = False
        assert np.array_equal(msk[b, keep], lab[b, keep])                 # untouched outside the index list
        pool = set(lab[b, :L][lab[b, :L] != 186].tolist())
        iset = set(idx.tolist())
        for (s, e) in _words(lab[b], L):
            n_words += 1
            inside = [i in iset for i in range(s, e)]
            assert all(inside) or not any(inside)                         # whole words
            if inside[0]:
                n_sel += 1
                if (msk[b, s:e] == 185).all():
                    n_mask += 1
                else:
                    n_repl += 1
                    assert set(msk[b, s:e].tolist()) <= pool              # replacements come from the sample
    # selected AND changed = 0.15 * 0.9 of the words; of those 8/9 masked, 1/9 replaced (binomial 4 sigma)
    p_idx = 0.15 * 0.9
    assert abs(n_sel / n_words - p_idx) < 4 * np.sqrt(p_idx * (1 - p_idx) / n_words)
    assert abs(n_mask / n_sel - 8 / 9) < 4 * np.sqrt((8 / 9) * (1 / 9) / n_sel)
    assert n_repl > 0


def test_reproducible_and_step_dependent_and_trains():
    lab, lengths = _labels(8, 128, 9)
    a = plbert_amd.device_mask_batch(lab, lengths, seed=3, step=1)
    b = plbert_amd.device_mask_batch(lab, lengths, seed=3, step=1)
    c = plbert_amd.device_mask_batch(lab, lengths, seed=3, step=2)
    assert torch.equal(a.masked, b.masked) and torch.equal(a.flat, b.flat)
    assert not torch.equal(a.masked, c.masked)
    cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=128, hidden_size=128, num_attention_heads=2,
                                  intermediate_size=256, num_hidden_layers=2)
    tr = plbert_amd.PLBertTrainer(cfg, 188, max_batch=8, max_seq=128, lr=1e-3)
    l0 = float(tr.step(a).item())
    for _ in range(5):
        l1 = float(tr.step(a).item())
    assert np.isfinite(l0) and l1 < l0
