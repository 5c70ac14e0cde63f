"""Bit-exact masking THROUGH the HIP path (-m gpu): the host draws the reference's random streams in the
reference's order (MaskedPhonemeDataset.decisions), plb_apply_mask does the integer work on the device; the result
must equal, bit for bit, the vectors captured from the reference's dataloader.py (tests/golden/masking.npz):
3-tuple and 4-tuple collated batches, max_seq_length 512 (no crop) and 32 (crop + index re-basing)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import plbert_amd
from plbert_amd import data as pdata
from plbert_amd.train import device_apply_mask

pytestmark = pytest.mark.gpu
PARAMS = dict(word_separator=87, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1)


def _docs(g):
    return [d.split("\x1f") for d in g["docs"]]


def _csr_lists(batch, B):
    off = batch.offsets.cpu().numpy()
    flat = batch.flat.cpu().numpy()
    return [flat[off[b]:off[b + 1]].tolist() for b in range(B)]


@pytest.mark.parametrize("tag,msl", [("msl512", 512), ("msl32", 32)])
def test_apply_mask_3tuple_equals_reference(tag, msl):
    g = load_golden("masking")
    data = [{"phonemes": d} for d in _docs(g)]
    pdata.seed_reference_streams(1)
    ds = plbert_amd.MaskedPhonemeDataset(data, max_seq_length=msl, use_token_ids=False, **PARAMS)
    recs = [ds.decisions(int(i)) for i in g[f"{tag}_order"]]          # consumes the streams exactly as __getitem__
    batch, labels, masked, lengths, tokens = device_apply_mask(recs[:8])
    torch.cuda.synchronize()
    assert tokens is None
    assert labels.dtype == torch.int64 and np.array_equal(labels.cpu().numpy(), g[f"{tag}_c3_labels"])
    assert np.array_equal(masked.cpu().numpy(), g[f"{tag}_c3_masked"])
    assert lengths == g[f"{tag}_c3_lengths"].tolist()
    got = _csr_lists(batch, 8)
    want = [np.asarray(x).tolist() for x in g[f"{tag}_c3_index"]]
    assert got == want
    assert batch.n_masked == sum(len(x) for x in want)
    # every item of the stream, one per batch: labels / masked / index of each __getitem__ call
    for k, r in enumerate(recs):
        b1, lab, msk, lens, _ = device_apply_mask([r])
        assert np.array_equal(lab.cpu().numpy()[0], g[f"{tag}_labels"][k])
        assert np.array_equal(msk.cpu().numpy()[0], g[f"{tag}_masked"][k])
        assert _csr_lists(b1, 1)[0] == np.asarray(g[f"{tag}_index"][k]).tolist()


@pytest.mark.parametrize("tag,msl", [("msl512", 512), ("msl32", 32)])
def test_apply_mask_4tuple_equals_reference(tag, msl):
    g = load_golden("masking")
    data = [{"phonemes": d, "token_ids": t.tolist()} for d, t in zip(_docs(g), g["token_ids"])]
    pdata.seed_reference_streams(1)
    ds = plbert_amd.MaskedPhonemeDataset(data, max_seq_length=msl, use_token_ids=True, **PARAMS)
    recs = [ds.decisions(int(i)) for i in g[f"{tag}_order"][:8]]
    batch, labels, masked, lengths, tokens = device_apply_mask(recs, word_separator=PARAMS["word_separator"])
    torch.cuda.synchronize()
    assert np.array_equal(tokens.cpu().numpy(), g[f"{tag}_c4_tokens"])
    assert np.array_equal(labels.cpu().numpy(), g[f"{tag}_c4_labels"])
    assert np.array_equal(masked.cpu().numpy(), g[f"{tag}_c4_masked"])
    assert lengths == g[f"{tag}_c4_lengths"].tolist()
    assert _csr_lists(batch, 8) == [np.asarray(x).tolist() for x in g[f"{tag}_c4_index"]]


def test_apply_mask_feeds_the_step_and_matches_host_path():
    """The device-built batch trains: same loss as the host-collated batch of the same decisions."""
    from plbert_amd.train import PLBertTrainer
    g = load_golden("masking")
    data = [{"phonemes": d} for d in _docs(g)]
    cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                  intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    pdata.seed_reference_streams(1)
    ds = plbert_amd.MaskedPhonemeDataset(data, max_seq_length=64, use_token_ids=False, **PARAMS)
    items = [ds[i] for i in (2, 4, 5, 7)]
    pdata.seed_reference_streams(1)
    recs = [ds.decisions(i) for i in (2, 4, 5, 7)]
    lab, msk, lens, idx = plbert_amd.PhonemeOnlyCollater()(items)
    tr = PLBertTrainer(cfg, 188, max_batch=4, max_seq=64, seed=1)
    l_host = float(tr.loss_and_grads(tr.stage_batch(lab.numpy(), msk.numpy(), lens, idx)).item())
    dev_batch, *_ = device_apply_mask(recs)
    l_dev = float(tr.loss_and_grads(dev_batch).item())
    assert l_host == l_dev
