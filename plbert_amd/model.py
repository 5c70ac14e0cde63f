"""Host-side mirror of the reference model interface, backed by the HIP engine.

Same names, constructor arguments and call conventions as the reference so its callers keep
working:

* ``AlbertModel(AlbertConfig(vocab_size=len(symbols), **config['model_params']))``  (train.py:263-265)
* ``PhonemeOnlyModel(model, num_phonemes, hidden_size)``                            (model.py:19-30)
* ``MultiTaskModel(model, num_phonemes, num_tokens, hidden_size)``                  (model.py:5-18)
* ``model(phonemes, attention_mask=(~text_mask).int())``                             (train.py:386)
* ``encoder(ids, attention_mask=...).last_hidden_state``                             (README.md:91)
* ``.parameters()`` / ``.state_dict()`` / ``.load_state_dict(strict=False)`` with the reference key
  names, ``.train()`` / ``.eval()``                                                 (train.py:100,272,290,417)

The modules are real ``torch.nn.Module`` trees whose Parameters are VIEWS into the engine's flat fp32
buffer, so ``load_state_dict``/``state_dict``/``torch.save`` behave as usual while the kernels see
one contiguous allocation.  Forward runs entirely in libplbert_hip.so (no autograd graph: outputs
are plain tensors; training goes through ``loss_and_grads`` / ``plbert_amd.train``).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from .engine import HipEngine
from .init import reference_init_state_dict

_DEFAULT_MAX_BATCH = 32


@dataclass
class BaseModelOutputWithPooling:
    last_hidden_state: torch.Tensor
    pooler_output: torch.Tensor | None = None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


class _Leaf(nn.Module):
    """Holds ``weight`` (and ``bias``) Parameters under the reference's attribute names."""

    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=True)
        if bias is not None:
            self.bias = nn.Parameter(bias, requires_grad=True)


def _lengths_from_mask(attention_mask, strict=True):
    """int [B,S] with 1 = valid -> int32 lengths. The kernels mask keys by length, so the mask must be
    a prefix mask (what train.py:384-386 builds from input_lengths)."""
    if attention_mask is None:
        return None
    am = attention_mask != 0
    lengths = am.sum(dim=1).to(torch.int32)
    if strict:
        S = am.shape[1]
        prefix = torch.arange(S, device=am.device)[None, :] < lengths[:, None]
        if not bool((prefix == am).all()):
            raise ValueError("attention_mask must mark a prefix of each row (1...1 0...0): the HIP attention "
                             "kernels take per-sample lengths")
        if bool((lengths < 1).any()):
            raise ValueError("attention_mask has a row with no valid token")
    return lengths


class AlbertModel(nn.Module):
    """Drop-in for ``transformers.AlbertModel`` on this path (modeling_albert.py:338-408)."""

    def __init__(self, config, add_pooling_layer=True, max_batch=_DEFAULT_MAX_BATCH, max_seq=None, device=None, seed=0):
        super().__init__()
        config.check_supported()
        self.config = config
        self._max_batch = max_batch
        self._max_seq = max_seq or config.max_position_embeddings
        self._device = device
        self._seed = seed
        self._engine = None
        # An encoder on its own is an inference model (README.md:91: bert(texts, attention_mask=...)): its engine holds
        # the parameters and, from the first forward on, one layer of activations (< 1 GB at 32 x 512) — no gradient,
        # moment or per-layer stash. Wrapped by PhonemeOnlyModel / MultiTaskModel its parameters move into the
        # wrapper's training engine before this one has allocated anything but them.
        self._build(HipEngine(config, 4, 0, max_batch=self._max_batch, max_seq=self._max_seq, device=device, train=False),
                    reference_init_state_dict(config, 4, 0, seed=seed), prefix="encoder.")

    # -- module tree with the reference's parameter names ----------------------------------------------
    def _build(self, engine, init_sd, prefix):
        """(Re)create the Parameter views over ``engine``'s flat buffer."""
        v = lambda name: engine.view(prefix + name)
        self._engine = engine
        if init_sd is not None:
            engine.load_state_dict({k: t for k, t in init_sd.items() if k in engine.layout}, strict=False)
        emb = nn.Module()
        emb.word_embeddings = _Leaf(v("embeddings.word_embeddings.weight"))
        emb.position_embeddings = _Leaf(v("embeddings.position_embeddings.weight"))
        emb.token_type_embeddings = _Leaf(v("embeddings.token_type_embeddings.weight"))
        emb.LayerNorm = _Leaf(v("embeddings.LayerNorm.weight"), v("embeddings.LayerNorm.bias"))
        self.embeddings = emb
        lp = "encoder.albert_layer_groups.0.albert_layers.0."
        layer = nn.Module()
        layer.full_layer_layer_norm = _Leaf(v(lp + "full_layer_layer_norm.weight"), v(lp + "full_layer_layer_norm.bias"))
        att = nn.Module()
        for nm in ("query", "key", "value", "dense", "LayerNorm"):
            setattr(att, nm, _Leaf(v(lp + f"attention.{nm}.weight"), v(lp + f"attention.{nm}.bias")))
        layer.attention = att
        layer.ffn = _Leaf(v(lp + "ffn.weight"), v(lp + "ffn.bias"))
        layer.ffn_output = _Leaf(v(lp + "ffn_output.weight"), v(lp + "ffn_output.bias"))
        group = nn.Module()
        group.albert_layers = nn.ModuleList([layer])
        enc = nn.Module()
        enc.embedding_hidden_mapping_in = _Leaf(v("encoder.embedding_hidden_mapping_in.weight"),
                                                v("encoder.embedding_hidden_mapping_in.bias"))
        enc.albert_layer_groups = nn.ModuleList([group])
        self.encoder = enc
        self.pooler = _Leaf(v("pooler.weight"), v("pooler.bias"))

    def _adopt(self, engine, owner):
        """Called by a head wrapper: move this encoder's values into the wrapper's engine."""
        old = {"encoder." + k: p.detach().clone() for k, p in self.state_dict().items()}
        self._build(engine, None, prefix="encoder.")
        engine.load_state_dict(old, strict=False)

    @property
    def engine(self):
        return self._engine

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, **unused):
        if input_ids is None:
            raise ValueError("input_ids is required (inputs_embeds is not supported on the HIP path)")
        if token_type_ids is not None or position_ids is not None:
            raise ValueError("the HIP path implements the reference's implicit token_type_ids=0 / position_ids=arange")
        lengths = _lengths_from_mask(attention_mask)
        hid, _, _ = self._engine.forward(input_ids, lengths, want_hidden=True, want_phoneme=False)
        # pooler (modeling_albert.py:403): computed for API completeness (plb_pooler), never used by the loss
        return BaseModelOutputWithPooling(last_hidden_state=hid, pooler_output=self._engine.pooler(hid))


class _HeadModel(nn.Module):
    def __init__(self, model, num_phonemes, num_tokens, hidden_size):
        super().__init__()
        if not isinstance(model, AlbertModel):
            raise TypeError("model must be a plbert_amd.AlbertModel")
        if hidden_size != model.config.hidden_size:
            raise ValueError("hidden_size does not match the encoder")
        cfg = model.config
        engine = HipEngine(cfg, num_phonemes, num_tokens, max_batch=model._max_batch, max_seq=model._max_seq,
                           device=model._device)
        init = reference_init_state_dict(cfg, num_phonemes, num_tokens, seed=model._seed)
        engine.load_state_dict({k: v for k, v in init.items() if not k.startswith("encoder.")}, strict=False)
        model._adopt(engine, self)
        self.encoder = model
        self.phoneme_predictor = _Leaf(engine.view("phoneme_predictor.weight"), engine.view("phoneme_predictor.bias"))
        if num_tokens:
            self.token_predictor = _Leaf(engine.view("token_predictor.weight"), engine.view("token_predictor.bias"))
        self._engine = engine

    @property
    def engine(self):
        return self._engine


class PhonemeOnlyModel(_HeadModel):
    """model.py:19-30 — ``forward(phonemes, attention_mask=None) -> phoneme_pred`` (fp32 [B,S,num_phonemes])."""

    def __init__(self, model, num_phonemes, hidden_size):
        super().__init__(model, num_phonemes, 0, hidden_size)

    def forward(self, phonemes, attention_mask=None):
        _, ph, _ = self._engine.forward(phonemes, _lengths_from_mask(attention_mask))
        return ph


class MultiTaskModel(_HeadModel):
    """model.py:5-18 — ``forward(phonemes, attention_mask=None) -> (phoneme_pred, token_pred)``."""

    def __init__(self, model, num_phonemes, num_tokens, hidden_size):
        super().__init__(model, num_phonemes, num_tokens, hidden_size)

    def forward(self, phonemes, attention_mask=None):
        _, ph, tk = self._engine.forward(phonemes, _lengths_from_mask(attention_mask), want_token=True)
        return ph, tk
