"""Configuration surface of the hot path: ``configs/config.yml`` + the AlbertConfig subset.

Mirrors how the reference builds its model (train.py:263-270):
``AlbertConfig(vocab_size=len(symbols), **config['model_params'])`` — unknown keys such as
``pretrained_model`` and ``dropout`` are accepted and kept as inert attributes.  Defaults not in
config.yml are HuggingFace's (configuration_albert.py:56-75): embedding_size 128, gelu_new,
dropout 0, layer_norm_eps 1e-12, type_vocab_size 2, one hidden group, one inner layer.
"""
from __future__ import annotations

import yaml


class AlbertConfig:
    def __init__(self, vocab_size=30000, embedding_size=128, hidden_size=4096, num_hidden_layers=12,
                 num_hidden_groups=1, num_attention_heads=64, intermediate_size=16384, inner_group_num=1,
                 hidden_act="gelu_new", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02,
                 layer_norm_eps=1e-12, pad_token_id=0, **extra):
        self.vocab_size = int(vocab_size)
        self.embedding_size = int(embedding_size)
        self.hidden_size = int(hidden_size)
        self.num_hidden_layers = int(num_hidden_layers)
        self.num_hidden_groups = int(num_hidden_groups)
        self.num_attention_heads = int(num_attention_heads)
        self.intermediate_size = int(intermediate_size)
        self.inner_group_num = int(inner_group_num)
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = float(hidden_dropout_prob)
        self.attention_probs_dropout_prob = float(attention_probs_dropout_prob)
        self.max_position_embeddings = int(max_position_embeddings)
        self.type_vocab_size = int(type_vocab_size)
        self.initializer_range = float(initializer_range)
        self.layer_norm_eps = float(layer_norm_eps)
        self.pad_token_id = pad_token_id
        for k, v in extra.items():  # pretrained_model, dropout, ... (train.py:263 splats them in)
            setattr(self, k, v)

    def check_supported(self):
        """What the HIP path supports; anything else fails loudly rather than falling back."""
        if self.hidden_act != "gelu_new":
            raise ValueError(f"hidden_act={self.hidden_act!r}: the HIP path implements gelu_new only")
        if self.num_hidden_groups != 1 or self.inner_group_num != 1:
            raise ValueError("only one shared layer (num_hidden_groups=1, inner_group_num=1) is implemented")
        if self.hidden_dropout_prob != 0.0 or self.attention_probs_dropout_prob != 0.0:
            raise ValueError("dropout must be 0 (the reference trains with HF's default 0.0)")
        if self.hidden_size % self.num_attention_heads != 0:
            raise ValueError("hidden_size must be a multiple of num_attention_heads")
        if self.hidden_size // self.num_attention_heads != 64:
            raise ValueError(f"head_dim = hidden_size / num_attention_heads = {self.hidden_size} / {self.num_attention_heads} = "
                             f"{self.hidden_size // self.num_attention_heads}: the attention kernels (csrc/attn.hip) keep a "
                             "head's 64 features in one MFMA operand and are built for head_dim 64 only "
                             "(configs/config.yml: 768 / 12; ALBERT-large: 1024 / 16)")
        # the same limits plb_create enforces (csrc/engine.cpp), reported here — when the model is constructed, not at its
        # first launch — with the reason
        if self.embedding_size % 64 or self.embedding_size > 256:
            raise ValueError(f"embedding_size={self.embedding_size}: must be a multiple of 64, at most 256 (the embedding "
                             "LayerNorm kernel holds a row in one wave)")
        if self.hidden_size % 128 or self.hidden_size > 1024:
            raise ValueError(f"hidden_size={self.hidden_size}: must be a multiple of 128, at most 1024 (a LayerNorm row spans "
                             "at most 4 column tiles of the fused GEMM epilogues)")
        if self.intermediate_size % 128:
            raise ValueError(f"intermediate_size={self.intermediate_size}: must be a multiple of 128 (GEMM tile width)")

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    def to_dict(self):
        return dict(self.__dict__)


def load_config(path):
    """Read a reference-format YAML (configs/config.yml)."""
    with open(path) as f:
        return yaml.safe_load(f)


def albert_config_from_yaml(config, vocab_size):
    """train.py:263 — AlbertConfig(vocab_size=len(symbols), **config['model_params'])."""
    return AlbertConfig(vocab_size=vocab_size, **config["model_params"])
