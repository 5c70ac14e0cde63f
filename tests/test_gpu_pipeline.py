"""Host -> device input pipeline (-m gpu): DeviceFeeder over a (multi-worker) DataLoader must hand
the step exactly the batches the plain host path collates — for collated tensors and for decision records whose
masking is applied on the GPU — while the copies run on their own stream."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import plbert_amd
from plbert_amd import data as pdata
from plbert_amd.pipeline import DeviceFeeder
from plbert_amd.train import PLBertTrainer

pytestmark = pytest.mark.gpu
PARAMS = dict(word_separator=87, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1)


def _docs(mult=6):
    g = load_golden("masking")
    return [{"phonemes": d.split("\x1f")} for d in g["docs"] if len(d) > 0] * mult


def _loaders(decisions, workers, seed=3):
    torch.manual_seed(seed)
    pdata.seed_reference_streams(1)
    return plbert_amd.build_dataloader(_docs(), batch_size=4, device="cpu", dataset_config=dict(max_seq_length=96, **PARAMS),
                                       use_token_ids=False, num_workers=workers, decisions=decisions)


def _csr(b):
    off, flat = b.offsets.cpu().numpy(), b.flat.cpu().numpy()
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


@pytest.mark.parametrize("workers", [0, 2])
def test_feeder_yields_the_host_batches(workers):
    want = [(lab.numpy().copy(), msk.numpy().copy(), list(lens), [list(x) for x in idx])
            for lab, msk, lens, idx in _loaders(False, workers)[0]]
    got = []
    for b in DeviceFeeder(_loaders(False, workers)[0], vocab_size=188):
        got.append((b.labels.cpu().numpy().copy(), b.masked.cpu().numpy().copy(), _csr(b), b.n_masked, b.n_tokens))
    assert len(got) == len(want) > 2
    for (lab, msk, idx, n, ntok), (wl, wm, wlens, widx) in zip(got, want):
        assert np.array_equal(lab, wl) and np.array_equal(msk, wm) and idx == widx
        assert n == sum(len(x) for x in widx) and ntok == sum(wlens)


def test_feeder_decisions_mode_equals_host_masking_and_trains():
    """Same decisions, applied on the GPU (plb_apply_mask) vs on the host: identical batches; and the batches feed the
    step (loss decreases over the epoch on repeated documents)."""
    want = [(lab.numpy().copy(), msk.numpy().copy(), [list(x) for x in idx], list(lens))
            for lab, msk, lens, idx in _loaders(False, 0)[0]]
    cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                  intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    tr = PLBertTrainer(cfg, 188, max_batch=4, max_seq=96, lr=1e-3, seed=1)
    ref = PLBertTrainer(cfg, 188, max_batch=4, max_seq=96, lr=1e-3, seed=1)
    k = 0
    for b in DeviceFeeder(_loaders(True, 0)[0], word_separator=PARAMS["word_separator"]):
        wl, wm, widx, wlens = want[k]
        assert np.array_equal(b.labels.cpu().numpy(), wl) and np.array_equal(b.masked.cpu().numpy(), wm)
        assert _csr(b) == widx
        # the device-built batch drives the same training trajectory as the host-collated one, bit for bit
        assert float(tr.step(b).item()) == float(ref.step(ref.stage_batch(wl, wm, wlens, widx)).item())
        k += 1
    assert k == len(want)
    assert torch.equal(tr.engine.params, ref.engine.params)
