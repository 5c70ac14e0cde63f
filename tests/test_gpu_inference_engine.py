"""Loss-only and inference-only entry points (-m gpu): validate() semantics (train.py:288-304) and the StyleTTS
consumer's encoder call (README.md:91) without paying for a training engine."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
import plbert_amd
from plbert_amd.engine import HipEngine
from plbert_amd.train import process_batch

pytestmark = pytest.mark.gpu


def _inputs(g):
    idx = [list(map(int, x)) for x in g["index"]]
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    return g["masked"], g["labels"], g["lengths"].astype(np.int32), off, flat, int(off[-1])


def test_loss_fwd_equals_loss_of_training_call_and_leaves_grads_alone():
    g = load_golden("small_h128")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    eng = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    args = _inputs(g)
    l_train = float(eng.loss_fwd_bwd(*args).item())
    sentinel = torch.full_like(eng.grads, 3.25)
    eng.grads.copy_(sentinel)
    l_val = float(eng.loss_fwd(*args).item())
    torch.cuda.synchronize()
    assert l_val == l_train                                    # same kernels, same order: bit-identical loss
    assert abs(l_val - float(g["loss"])) / float(g["loss"]) < 1e-3
    assert torch.equal(eng.grads, sentinel)                    # validate() must not clobber the gradient buffer
    # an inference-only engine gives the same loss and refuses to train
    inf = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S, train=False)
    inf.load_state_dict(sd)
    assert float(inf.loss_fwd(*args).item()) == l_val
    assert inf.grads is None and inf.exp_avg is None
    with pytest.raises(RuntimeError):
        inf.loss_fwd_bwd(*args)
    assert inf.device_bytes() < 0.25 * eng.device_bytes()


def test_process_batch_under_no_grad_is_validation():
    g = load_golden("small_h128")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    enc = plbert_amd.AlbertModel(pcfg, max_batch=B, max_seq=S)
    model = plbert_amd.PhonemeOnlyModel(enc, 188, pcfg.hidden_size)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    idx = [list(map(int, x)) for x in g["index"]]
    batch = (torch.from_numpy(g["labels"]), torch.from_numpy(g["masked"]), [int(x) for x in g["lengths"]], idx)
    loss = process_batch(model, batch)
    loss.backward()
    before = model.engine.grads.clone()
    model.eval()
    with torch.no_grad():
        lv = process_batch(model, batch)
    torch.cuda.synchronize()
    assert abs(float(lv) - float(g["loss"])) / float(g["loss"]) < 1e-3
    assert torch.equal(model.engine.grads, before)
    assert enc.engine is model.engine                          # one engine per model after wrapping


def test_encoder_alone_is_cheap_and_correct():
    """AlbertModel at the bench capacity (32 x 512, 768/12) holds < 1 GB; its hidden states equal the training
    engine's (same kernels, one layer of activations reused)."""
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                  max_position_embeddings=512, num_hidden_layers=12)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    m0 = torch.cuda.memory_allocated()
    enc = plbert_amd.AlbertModel(cfg, max_batch=32, max_seq=512)
    assert torch.cuda.memory_allocated() - m0 < 64 << 20       # constructed: parameters only
    ids = torch.from_numpy(np.random.RandomState(0).randint(1, 185, size=(3, 200))).cuda()
    mask = torch.ones(3, 200, dtype=torch.int32, device="cuda")
    mask[1, 150:] = 0
    out = enc(ids, attention_mask=mask)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() - m0 < 1 << 30
    assert enc.engine.device_bytes() < 1 << 30
    assert out.last_hidden_state.shape == (3, 200, 768) and out.pooler_output.shape == (3, 768)
    pooled = torch.tanh(torch.nn.functional.linear(out.last_hidden_state[:, 0], enc.pooler.weight, enc.pooler.bias))
    assert float((out.pooler_output - pooled).abs().max()) < 1e-5
    ref = HipEngine(cfg, 4, 0, max_batch=3, max_seq=200)
    ref.load_state_dict({"encoder." + k: v for k, v in enc.state_dict().items()}, strict=False)
    hid, _, _ = ref.forward(ids, torch.tensor([200, 150, 200], dtype=torch.int32), want_hidden=True, want_phoneme=False)
    torch.cuda.synchronize()
    valid = mask.bool()
    assert torch.equal(hid[valid], out.last_hidden_state[valid])


def test_engine_on_a_named_device_while_another_context_is_current():
    """every C-ABI call runs under the engine's device guard."""
    cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                  intermediate_size=256, num_hidden_layers=1, max_position_embeddings=64)
    eng = HipEngine(cfg, 188, 0, max_batch=1, max_seq=16, device="cuda:0")
    assert eng.device == torch.device("cuda", 0)
    _, ph, _ = eng.forward(np.ones((1, 16), dtype=np.int64))
    torch.cuda.synchronize()
    assert torch.isfinite(ph).all()
