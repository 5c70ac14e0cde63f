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
