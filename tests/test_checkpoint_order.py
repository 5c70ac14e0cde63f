"""The optimizer entry of a ``step_N.pth`` (train.py:416-421) indexes its state by position in the reference's
``model.parameters()`` — HF ``AlbertModel``'s module order, weight before bias, pooler before the heads — NOT by the
engine's flat layout (q.w k.w v.w q.b k.b v.b …, phoneme head before the pooler). CPU tests of the mapping that
``plbert_amd.run`` uses in both directions, against ``torch.optim.AdamW`` over the reference's model structure."""
import numpy as np
import pytest
import torch
from torch import nn

import plbert_amd
from plbert_amd import checkpoint as ck


def _reference_model(num_tokens=0):
    """model.py:5-30 restated over the installed transformers.AlbertModel (the module order is what matters here)."""
    transformers = pytest.importorskip("transformers")
    hf = transformers.AlbertConfig(vocab_size=40, embedding_size=16, hidden_size=32, num_attention_heads=2,
                                   intermediate_size=48, num_hidden_layers=2, max_position_embeddings=24,
                                   hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    class Heads(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = transformers.AlbertModel(hf)
            self.phoneme_predictor = nn.Linear(32, 40)
            if num_tokens:
                self.token_predictor = nn.Linear(32, num_tokens)

    cfg = plbert_amd.AlbertConfig(vocab_size=40, embedding_size=16, hidden_size=32, num_attention_heads=2,
                                  intermediate_size=48, num_hidden_layers=2, max_position_embeddings=24)
    return Heads(), cfg


def _flat_layout(cfg, num_phonemes, num_tokens=0):
    layout, off = {}, 0
    for n, shp in plbert_amd.param_shapes(cfg, num_phonemes, num_tokens).items():
        size = int(np.prod(shp))
        layout[n] = (off, size, tuple(shp))
        off += size
    return layout, off


@pytest.mark.parametrize("num_tokens", [0, 50])
def test_reference_order_is_torch_parameters_order(num_tokens):
    model, cfg = _reference_model(num_tokens)
    layout, _ = _flat_layout(cfg, 40, num_tokens)
    names = [n for n, _ in model.named_parameters()]
    assert ck.reference_param_names(layout) == names
    assert list(layout) != names                       # the engine's order really is a different one


def test_torch_adamw_state_round_trips_by_name():
    torch.manual_seed(0)
    model, cfg = _reference_model()
    layout, total = _flat_layout(cfg, 40)
    params = dict(model.named_parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    for n, p in params.items():
        if "pooler" not in n:                          # the reference's step never reaches the pooler (train.py:383-390)
            p.grad = torch.randn_like(p)
    opt.step()
    sd = opt.state_dict()
    names = [n for n, _ in model.named_parameters()]
    exp_avg, exp_avg_sq = torch.full((total,), 7.0), torch.full((total,), 7.0)
    steps = ck.optimizer_state_to_flat(sd["state"], layout, exp_avg, exp_avg_sq, lambda n, off, size: "pooler" not in n)
    assert set(steps.values()) == {1} and not any("pooler" in n for n in steps)
    for i, n in enumerate(names):                      # every moment sits under ITS tensor's offset
        off, size, shp = layout[n]
        want = sd["state"][i]["exp_avg"] if i in sd["state"] else torch.zeros(shp)
        assert torch.equal(exp_avg[off:off + size].view(shp), want), n
    # and back: the file this build writes is accepted by torch's AdamW over the reference model, tensor for tensor
    state, n_params = ck.optimizer_state_from_flat(layout, exp_avg, exp_avg_sq, lambda n, off, size: steps.get(n, 0))
    assert n_params == len(names)
    opt2 = torch.optim.AdamW(model.parameters(), lr=1e-3)
    group = dict(sd["param_groups"][0], params=list(range(n_params)))
    opt2.load_state_dict({"state": state, "param_groups": [group]})
    for n, p in params.items():
        if "pooler" in n:
            assert p not in opt2.state
        else:
            assert torch.equal(opt2.state[p]["exp_avg"], opt.state[p]["exp_avg"]), n
            assert torch.equal(opt2.state[p]["exp_avg_sq"], opt.state[p]["exp_avg_sq"]), n


def test_state_in_engine_order_is_refused_not_broadcast():
    """A file whose indices count the engine's flat order (what round 2 wrote) puts q.bias [H] at the index of
    key.weight [H, H]: ``copy_`` would broadcast it silently; the loader must raise instead."""
    model, cfg = _reference_model()
    layout, total = _flat_layout(cfg, 40)
    state = {i: {"step": torch.tensor(1.0), "exp_avg": torch.zeros(shp), "exp_avg_sq": torch.zeros(shp)}
             for i, (n, (off, size, shp)) in enumerate(layout.items())}
    with pytest.raises(ValueError, match="shape"):
        ck.optimizer_state_to_flat(state, layout, torch.zeros(total), torch.zeros(total), lambda *a: True)
