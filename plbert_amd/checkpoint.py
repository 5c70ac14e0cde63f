"""Checkpoint files of the reference run layout (train.py:46-105, 412-425).

``step_N.pth`` = ``{'net': state_dict, 'step': N, 'epoch': E, 'optimizer': optimizer.state_dict()}``
written with ``torch.save``; keys may carry DDP's ``module.`` prefix, which loading strips
(train.py:98) before a ``strict=False`` load (train.py:100). The optimizer entry uses torch's AdamW
layout, so files move both ways between the reference and this implementation.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


_LAYER = "encoder.encoder.albert_layer_groups.0.albert_layers.0."
# ``PhonemeOnlyModel(AlbertModel(cfg), …).parameters()`` / ``MultiTaskModel(…).parameters()`` order of the reference
# (model.py:5-30 over HF modeling_albert.py's module order): each Linear / LayerNorm yields weight then bias, the
# encoder's pooler comes before the heads. torch.optim.AdamW.state_dict() indexes its state by position in THIS order —
# not by the engine's flat layout (q.w k.w v.w q.b k.b v.b …, phoneme head before the pooler), which exists so that the
# fused QKV operand and the trainable range are contiguous.
REFERENCE_PARAM_ORDER = (
    ["encoder.embeddings.word_embeddings.weight", "encoder.embeddings.position_embeddings.weight",
     "encoder.embeddings.token_type_embeddings.weight", "encoder.embeddings.LayerNorm.weight",
     "encoder.embeddings.LayerNorm.bias", "encoder.encoder.embedding_hidden_mapping_in.weight",
     "encoder.encoder.embedding_hidden_mapping_in.bias"]
    + [_LAYER + n + sfx for n in ("full_layer_layer_norm", "attention.query", "attention.key", "attention.value",
                                  "attention.dense", "attention.LayerNorm", "ffn", "ffn_output")
       for sfx in (".weight", ".bias")]
    + ["encoder.pooler.weight", "encoder.pooler.bias", "phoneme_predictor.weight", "phoneme_predictor.bias",
       "token_predictor.weight", "token_predictor.bias"])


def reference_param_names(layout):
    """The names of ``layout`` (the engine's ``{name: (offset, size, shape)}``) in the reference's ``parameters()``
    order: position i of the result is index i of a torch AdamW state dict."""
    names = [n for n in REFERENCE_PARAM_ORDER if n in layout]
    missing = set(layout) - set(names)
    if missing:
        raise KeyError(f"parameters without a place in the reference order: {sorted(missing)}")
    return names


def optimizer_state_from_flat(layout, exp_avg, exp_avg_sq, steps_of):
    """torch.optim.AdamW ``state`` ({index: {step, exp_avg, exp_avg_sq}}) from the flat moment buffers; ``steps_of(name,
    offset, size)`` gives a tensor's step count (0: never updated -> no entry, as in torch). Indices follow
    ``reference_param_names``."""
    names = reference_param_names(layout)
    state = {}
    for i, n in enumerate(names):
        off, size, shp = layout[n]
        steps = steps_of(n, off, size)
        if steps > 0:
            state[i] = {"step": torch.tensor(float(steps)), "exp_avg": exp_avg[off:off + size].view(shp).cpu().clone(),
                        "exp_avg_sq": exp_avg_sq[off:off + size].view(shp).cpu().clone()}
    return state, len(names)


def optimizer_state_to_flat(state, layout, exp_avg, exp_avg_sq, accepts):
    """Inverse of ``optimizer_state_from_flat``: copy every entry ``accepts(name, offset, size)`` allows into the flat
    buffers (zeroed first). Shapes must match exactly — ``copy_`` would broadcast a [H] bias into a [H, H] weight.
    Returns {name: step}."""
    names = reference_param_names(layout)
    exp_avg.zero_()
    exp_avg_sq.zero_()
    steps = {}
    for i, st in state.items():
        i = int(i)
        if not 0 <= i < len(names):
            raise IndexError(f"optimizer state index {i} outside the model's {len(names)} parameters")
        n = names[i]
        off, size, shp = layout[n]
        for key in ("exp_avg", "exp_avg_sq"):
            if tuple(st[key].shape) != tuple(shp):
                raise ValueError(f"optimizer state {i} ({n}): {key} has shape {tuple(st[key].shape)}, parameter {tuple(shp)}")
        if not accepts(n, off, size):
            continue
        exp_avg[off:off + size].view(shp).copy_(st["exp_avg"])
        exp_avg_sq[off:off + size].view(shp).copy_(st["exp_avg_sq"])
        steps[n] = int(float(st["step"]))
    return steps


def find_latest_checkpoint(log_dir):
    """(found, last_step): the largest N among regular files named ``step_N[.ext]`` (train.py:46-79)."""
    steps = []
    try:
        for f in os.listdir(log_dir):
            if f.startswith("step_") and os.path.isfile(os.path.join(log_dir, f)):
                try:
                    steps.append(int(f.split("_")[-1].split(".")[0]))
                except ValueError:
                    continue
    except OSError as ex:
        print(f"Error finding checkpoints: {ex}")
        return False, 0
    return (True, max(steps)) if steps else (False, 0)


def _is_main():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_checkpoint(model, optimizer, current_step, log_dir, accelerator=None, current_epoch=0):
    """train.py:412-425 — main process only; returns the path."""
    path = os.path.join(log_dir, f"step_{current_step}.pth")
    if _is_main():
        state = {"net": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "step": current_step,
                 "epoch": current_epoch,
                 "optimizer": torch.utils._pytree.tree_map(lambda t: t.cpu() if torch.is_tensor(t) else t,
                                                           optimizer.state_dict())}
        os.makedirs(log_dir, exist_ok=True)
        torch.save(state, path)
        print(f"Checkpoint saved at: {path}")
    return path


def load_checkpoint(model, optimizer, checkpoint_path, accelerator=None):
    """train.py:81-105 — returns (model, optimizer) with the loaded state."""
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    net = {k.replace("module.", ""): v for k, v in ck["net"].items()}
    model.load_state_dict(net, strict=False)
    if optimizer is not None and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    print(f"Checkpoint {checkpoint_path} loaded.")
    return model, optimizer
