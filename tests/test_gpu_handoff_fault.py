"""A timed-out in-launch hand-off of the fused LayerNorm GEMMs (csrc/gemm_nt_pipeline.h, forms 5 / 6) must be impossible to
miss and impossible to carry over (include/plbert.h: plb_status / plb_poll_status):

  * the loss of the failed step is NaN, the optimizer update of that step is skipped ON THE DEVICE (no host round trip),
  * the next engine call raises HandoffTimeout (one pinned-word read, no synchronisation) and rewinds the step count,
  * after the report the exchange buffer is clean again: the following step is bit-identical to an undisturbed run —
    also when the failed launch left TAGGED granules behind (what a producer's store landing after the time-out does).

The fault is injected through plb_debug_ln_fault (a test hook of the library, documented at the end of the header): no
real time-out has ever been observed (ln_exchange_timeouts is asserted 0 in the engine, large-batch and two-rank tests)."""
import math

import numpy as np
import pytest
import torch

import plbert_amd
from plbert_amd import _lib
from plbert_amd.engine import HandoffTimeout

pytestmark = pytest.mark.gpu


def _trainer():
    # 768-wide rows = two 384-column tiles per row block; 2 x 512 = 1024 tokens: the engine runs the fused forms
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                  max_position_embeddings=512, num_hidden_layers=2)
    tr = plbert_amd.PLBertTrainer(cfg, 188, max_batch=2, max_seq=512, seed=3, lr=1e-3)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(2, 512, seed=5)
    return tr, tr.stage_batch(labels, masked, lengths, idx)


def _run_clean(n):
    tr, b = _trainer()
    losses = [float(tr.step(b).item()) for _ in range(n)]
    assert tr.engine.status()["ln_exchange_timeouts"] == 0
    return losses, tr.engine.params.clone()


@pytest.mark.parametrize("mode", [1, 2])
def test_timed_out_handoff_raises_skips_the_update_and_leaves_nothing_behind(mode):
    ref_losses, ref_params = _run_clean(3)
    tr, b = _trainer()
    L = _lib.lib()
    assert float(tr.step(b).item()) == ref_losses[0]
    torch.cuda.synchronize()
    p1 = tr.engine.params.clone()
    try:
        L.plb_debug_ln_fault(mode, 1)                      # the next fused LayerNorm launch loses a hand-off
        bad = tr.step(b)                                   # returns as always: nothing on the host waits for the device
        torch.cuda.synchronize()
    finally:
        L.plb_debug_ln_fault(0, 0)
    assert math.isnan(float(bad.item()))                   # the loss says so ...
    assert torch.equal(tr.engine.params, p1)               # ... and the device left the update out
    assert tr.step_count == 2                              # (the host has not noticed yet)
    assert tr.engine.poll_status()["ln_exchange_timeouts"] > 0
    with pytest.raises(HandoffTimeout) as ei:
        tr.step(b)                                         # first thing the next step does: one read of a pinned word
    assert ei.value.skipped_updates == 1
    assert tr.step_count == 1                              # bias correction continues where the last applied update left it
    assert tr.engine.poll_status()["ln_exchange_timeouts"] == 0
    assert torch.equal(tr.engine.params, p1)
    # the engine has been reset: the run continues exactly as if nothing had happened
    assert float(tr.step(b).item()) == ref_losses[1]
    assert float(tr.step(b).item()) == ref_losses[2]
    torch.cuda.synchronize()
    assert tr.engine.status() == {"ln_exchange_timeouts": 0, "skipped_updates": 0}
    assert torch.equal(tr.engine.params, ref_params)


def test_failed_steps_stay_skipped_until_reported():
    """The error word is sticky: steps enqueued behind a failed one (the host runs ahead of the device) are skipped
    too, and the report says how many."""
    tr, b = _trainer()
    L = _lib.lib()
    tr.step(b)
    torch.cuda.synchronize()
    p1 = tr.engine.params.clone()
    try:
        L.plb_debug_ln_fault(1, 1)
        eng = tr.engine
        for _ in range(3):                                  # what a host that never reads anything back would enqueue
            eng.L.plb_loss_fwd_bwd(eng.handle, b.masked.data_ptr(), b.labels.data_ptr(), None, b.offsets.data_ptr(),
                                   b.flat.data_ptr(), b.n_masked, 2, 512, eng._loss.data_ptr(), eng._stream())
            eng.adamw_step(2, lr=1e-3)
        torch.cuda.synchronize()
    finally:
        L.plb_debug_ln_fault(0, 0)
    assert math.isnan(float(tr.engine._loss.item()))
    assert torch.equal(tr.engine.params, p1)
    st = tr.engine.status()
    assert st["ln_exchange_timeouts"] > 0 and st["skipped_updates"] == 3
    assert tr.engine.status() == {"ln_exchange_timeouts": 0, "skipped_updates": 0}   # reported once


def test_process_batch_raises_too():
    """The reference-shaped loop (process_batch + AdamW, train.py:352-357) goes through the same engine call."""
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                  max_position_embeddings=512, num_hidden_layers=2)
    model = plbert_amd.PhonemeOnlyModel(plbert_amd.AlbertModel(cfg), num_phonemes=188, hidden_size=768).cuda()
    opt = plbert_amd.AdamW(model.parameters(), lr=1e-3, model=model)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(2, 512, seed=5)
    batch = (torch.as_tensor(labels), torch.as_tensor(masked), lengths, idx)
    L = _lib.lib()

    def step():
        loss = plbert_amd.process_batch(model, batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    step()
    torch.cuda.synchronize()
    before = model.engine.params.clone()
    try:
        L.plb_debug_ln_fault(1, 1)
        bad = step()
        torch.cuda.synchronize()
    finally:
        L.plb_debug_ln_fault(0, 0)
    assert math.isnan(float(bad.item()))
    assert torch.equal(model.engine.params, before)
    with pytest.raises(HandoffTimeout):
        step()
    assert opt.step_count == 1
    assert np.isfinite(float(step().item()))


def test_dual_head_step_count_follows_the_skipped_updates():
    """The token head keeps an AdamW step count of its own (plb_token_head_steps, torch keeps one per parameter): an update
    the device left out must not advance it either, or the head's next update runs with the wrong bias correction."""
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                  max_position_embeddings=512, num_hidden_layers=2)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(2, 512, seed=5)
    tok = np.random.RandomState(9).randint(0, 512, size=(2, 512)).astype(np.int64)

    def make():
        tr = plbert_amd.PLBertTrainer(cfg, 188, max_batch=2, max_seq=512, seed=3, lr=1e-3, num_tokens=512)
        return tr, tr.stage_batch(labels, masked, lengths, idx, token_ids=tok)

    tr, b = make()
    ref = [float(tr.step(b).item()) for _ in range(3)]
    torch.cuda.synchronize()
    ref_params = tr.engine.params.clone()
    assert tr.engine.token_head_steps == 3
    del tr
    tr, b = make()
    L = _lib.lib()
    assert float(tr.step(b).item()) == ref[0]
    try:
        L.plb_debug_ln_fault(1, 1)
        bad = tr.step(b)
        torch.cuda.synchronize()
    finally:
        L.plb_debug_ln_fault(0, 0)
    assert math.isnan(float(bad.item()))
    assert tr.engine.token_head_steps == 2                 # the host counted the launch it enqueued ...
    with pytest.raises(HandoffTimeout):
        tr.step(b)
    assert tr.engine.token_head_steps == 1 and tr.step_count == 1   # ... and takes it back with the report
    assert float(tr.step(b).item()) == ref[1]
    assert float(tr.step(b).item()) == ref[2]
    torch.cuda.synchronize()
    assert tr.engine.token_head_steps == 3
    assert torch.equal(tr.engine.params, ref_params)
