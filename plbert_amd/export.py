"""Hand a trained encoder to its consumers (second half).

Two consumers exist in the reference tree:

* the StyleTTS recipe (README.md:40-66) takes a ``step_N`` file, drops ``module.`` and keeps the keys under
  ``encoder.`` with that prefix removed, and loads the result into ``transformers.AlbertModel``:
  ``encoder_state_dict`` is that key transformation;
* ``convert_to_hf.py:16-64`` writes a directory — the encoder via ``save_pretrained`` (``config.json`` + weights),
  ``pl_bert_full_model.pt`` (the whole state dict with the heads), ``training_metadata.txt`` and ``config.yml`` —
  which ``load_pl_bert_model`` (convert_to_hf.py:66-102) reads back: ``export_pretrained`` writes the same directory
  from a checkpoint file or a state dict, without a tokenizer download (``num_tokens`` comes from the weights).

Host-side only: tensors in, files out. Nothing here touches the device library, so a checkpoint can be exported on a
machine without a GPU.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch
import yaml

from .config import AlbertConfig, albert_config_from_yaml
from .symbols import symbols

# keys of transformers' AlbertConfig that describe the encoder (configuration_albert.py:56-75); anything else in
# config.yml's model_params (pretrained_model, dropout) is the training script's business and stays in config.yml
_HF_KEYS = ("vocab_size", "embedding_size", "hidden_size", "num_hidden_layers", "num_hidden_groups",
            "num_attention_heads", "intermediate_size", "inner_group_num", "hidden_act", "hidden_dropout_prob",
            "attention_probs_dropout_prob", "max_position_embeddings", "type_vocab_size", "initializer_range",
            "layer_norm_eps", "pad_token_id")


def strip_module(state_dict):
    """DDP's prefix (train.py:98, convert_to_hf.py:40)."""
    return OrderedDict((k.replace("module.", ""), v) for k, v in state_dict.items())


def encoder_state_dict(net_state_dict):
    """README.md:57-63 — the ``encoder.*`` entries of a checkpoint's ``net``, as ``AlbertModel`` names them."""
    out = OrderedDict()
    for k, v in strip_module(net_state_dict).items():
        if k.startswith("encoder."):
            out[k[len("encoder."):]] = v
    if not out:
        raise ValueError("no 'encoder.*' entries: not a PhonemeOnlyModel / MultiTaskModel state dict")
    return out


def hf_config_dict(cfg: AlbertConfig):
    d = {k: getattr(cfg, k) for k in _HF_KEYS}
    d.update(model_type="albert", architectures=["AlbertModel"], classifier_dropout_prob=0.1,
             bos_token_id=2, eos_token_id=3, dtype="float32")
    return d


def _check_shapes(enc, cfg):
    want = {
        "embeddings.word_embeddings.weight": (cfg.vocab_size, cfg.embedding_size),
        "encoder.embedding_hidden_mapping_in.weight": (cfg.hidden_size, cfg.embedding_size),
        "encoder.albert_layer_groups.0.albert_layers.0.ffn.weight": (cfg.intermediate_size, cfg.hidden_size),
    }
    for k, shp in want.items():
        if k not in enc:
            raise ValueError(f"state dict has no {k!r}")
        if tuple(enc[k].shape) != shp:
            raise ValueError(f"{k}: shape {tuple(enc[k].shape)} does not match the configuration {shp}")


def export_pretrained(source, config, output_dir, step=None, epoch=None):
    """Write convert_to_hf.py's directory. ``source``: path of a ``step_N.pth`` file, a checkpoint dict with
    ``'net'``, or a model state dict; ``config``: the YAML dict (``model_params`` required). Returns the paths."""
    from safetensors.torch import save_file
    origin = "(state dict)"
    if isinstance(source, (str, os.PathLike)):
        origin = os.fspath(source)
        source = torch.load(source, map_location="cpu", weights_only=False)
    if "net" in source and not torch.is_tensor(source["net"]):
        step = source.get("step", step) if step is None else step
        epoch = source.get("epoch", epoch) if epoch is None else epoch
        source = source["net"]
    full = OrderedDict((k, v.detach().to("cpu", torch.float32).contiguous()) for k, v in strip_module(source).items())
    cfg = albert_config_from_yaml(config, vocab_size=len(symbols))
    enc = encoder_state_dict(full)
    _check_shapes(enc, cfg)
    os.makedirs(output_dir, exist_ok=True)
    paths = {k: os.path.join(output_dir, v) for k, v in
             dict(config="config.json", weights="model.safetensors", full="pl_bert_full_model.pt",
                  metadata="training_metadata.txt", yaml="config.yml").items()}
    with open(paths["config"], "w") as f:
        json.dump(hf_config_dict(cfg), f, indent=2, sort_keys=True)
        f.write("\n")
    save_file({k: v.clone() for k, v in enc.items()}, paths["weights"], metadata={"format": "pt"})
    torch.save(full, paths["full"])
    with open(paths["metadata"], "w") as f:  # convert_to_hf.py:55-58
        f.write(f"Original checkpoint: {origin}\nStep: {step}\nEpoch: {epoch}\n")
    with open(paths["yaml"], "w") as f:
        yaml.dump(config, f)
    return paths


def read_exported(model_dir):
    """(config dict, full state dict, num_tokens or 0) of a directory written by ``export_pretrained`` or by the
    reference's converter — the host half of ``load_pl_bert_model``."""
    with open(os.path.join(model_dir, "config.yml")) as f:
        config = yaml.safe_load(f)
    full = strip_module(torch.load(os.path.join(model_dir, "pl_bert_full_model.pt"), map_location="cpu",
                                   weights_only=False))
    tok = full.get("token_predictor.weight")
    return config, full, (0 if tok is None else int(tok.shape[0]))


def load_pl_bert_model(model_dir, device=None, **engine_args):
    """convert_to_hf.py:66-102 on the native path: the model (``MultiTaskModel`` when the directory holds a token
    head, else ``PhonemeOnlyModel``) in eval mode. The tokenizer of the reference's return pair is a hub download and
    is not part of this path: ``num_tokens`` is read from the stored head."""
    from .model import AlbertModel, MultiTaskModel, PhonemeOnlyModel
    config, full, num_tokens = read_exported(model_dir)
    cfg = albert_config_from_yaml(config, vocab_size=len(symbols))
    bert = AlbertModel(cfg, device=device, **engine_args)
    H = cfg.hidden_size
    model = (MultiTaskModel(bert, len(symbols), num_tokens, H) if num_tokens
             else PhonemeOnlyModel(bert, len(symbols), H))
    missing = [k for k in model.state_dict() if k not in full]
    if missing:
        raise KeyError(f"{model_dir}: pl_bert_full_model.pt lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
    model.load_state_dict(full)
    model.eval()
    return model


def main(argv=None):
    """``python -m plbert_amd.export --checkpoint_path … --config_path … --output_dir …`` (convert_to_hf.py:9-14)."""
    import argparse
    ap = argparse.ArgumentParser(description="Export a PL-BERT checkpoint as an AlbertModel directory + full state dict")
    ap.add_argument("--checkpoint_path", required=True)
    ap.add_argument("--config_path", required=True)
    ap.add_argument("--output_dir", required=True)
    a = ap.parse_args(argv)
    with open(a.config_path) as f:
        config = yaml.safe_load(f)
    export_pretrained(a.checkpoint_path, config, a.output_dir)
    print(f"Model successfully converted and saved to {a.output_dir}")


if __name__ == "__main__":
    main()
