"""Parameter inventory and initialisers for the PL-BERT model (state-dict names of SURVEY.md §8(b)).

``param_shapes`` is the single source of the flat parameter layout the HIP engine uses.
``deterministic_state_dict`` is the generator the golden fixtures were captured with
(np.random.RandomState(seed).standard_normal * 0.02; LayerNorm weight 1 + noise, biases small and
non-zero so every bias path is exercised).  ``reference_init_state_dict`` mirrors the reference's
actual initialisation: HF ``_init_weights`` N(0, initializer_range) for the encoder, zeros for
biases, ones/zeros for LayerNorm, zero pad row; PyTorch default ``nn.Linear`` init for the heads
(model.py:10-11,24).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

ENC = "encoder."
LAYER = "encoder.encoder.albert_layer_groups.0.albert_layers.0."


def param_shapes(cfg, num_phonemes, num_tokens=0):
    V, E, H, I = cfg.vocab_size, cfg.embedding_size, cfg.hidden_size, cfg.intermediate_size
    s = OrderedDict()
    s[ENC + "embeddings.word_embeddings.weight"] = (V, E)
    s[ENC + "embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, E)
    s[ENC + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, E)
    s[ENC + "embeddings.LayerNorm.weight"] = (E,)
    s[ENC + "embeddings.LayerNorm.bias"] = (E,)
    s[ENC + "encoder.embedding_hidden_mapping_in.weight"] = (H, E)
    s[ENC + "encoder.embedding_hidden_mapping_in.bias"] = (H,)
    s[LAYER + "full_layer_layer_norm.weight"] = (H,)
    s[LAYER + "full_layer_layer_norm.bias"] = (H,)
    # query/key/value are adjacent so [3H, H] / [3H] views of the flat buffer are the fused QKV operand
    for nm in ("query", "key", "value"):
        s[LAYER + f"attention.{nm}.weight"] = (H, H)
    for nm in ("query", "key", "value"):
        s[LAYER + f"attention.{nm}.bias"] = (H,)
    s[LAYER + "attention.dense.weight"] = (H, H)
    s[LAYER + "attention.dense.bias"] = (H,)
    s[LAYER + "attention.LayerNorm.weight"] = (H,)
    s[LAYER + "attention.LayerNorm.bias"] = (H,)
    s[LAYER + "ffn.weight"] = (I, H)
    s[LAYER + "ffn.bias"] = (I,)
    s[LAYER + "ffn_output.weight"] = (H, I)
    s[LAYER + "ffn_output.bias"] = (H,)
    s["phoneme_predictor.weight"] = (num_phonemes, H)
    s["phoneme_predictor.bias"] = (num_phonemes,)
    # parameters below never receive a gradient in the reference's step (train.py:383-390)
    s[ENC + "pooler.weight"] = (H, H)
    s[ENC + "pooler.bias"] = (H,)
    if num_tokens:
        s["token_predictor.weight"] = (num_tokens, H)
        s["token_predictor.bias"] = (num_tokens,)
    return s


def deterministic_state_dict(cfg, num_phonemes, num_tokens=0, seed=0, scale=0.02):
    """Golden-fixture generator: draws in sorted-name order so the layout order never matters."""
    shapes = param_shapes(cfg, num_phonemes, num_tokens)
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        x = rs.standard_normal(shp).astype(np.float32) * np.float32(scale)
        if name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
            x = x + np.float32(1.0)
        out[name] = x
    return OrderedDict((n, out[n]) for n in shapes)


def reference_init_state_dict(cfg, num_phonemes, num_tokens=0, seed=0):
    shapes = param_shapes(cfg, num_phonemes, num_tokens)
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shp in shapes.items():
        head = name.startswith("phoneme_predictor") or name.startswith("token_predictor")
        if head:
            fan_in = cfg.hidden_size
            bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) on [out,in] == U(-1/sqrt(in), 1/sqrt(in))
            x = rs.uniform(-bound, bound, size=shp).astype(np.float32)
        elif name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
            x = np.ones(shp, np.float32)
        elif name.endswith(".bias"):
            x = np.zeros(shp, np.float32)
        else:
            x = (rs.standard_normal(shp) * cfg.initializer_range).astype(np.float32)
            if name.endswith("word_embeddings.weight") and cfg.pad_token_id is not None:
                x[cfg.pad_token_id] = 0.0
        out[name] = x
    return out
