"""Data-parallel plumbing of the hot path: one process per GPU, replicated
parameters, the global batch split by sample across ranks and ONE exchange step per training step —
a sum all-reduce of the flat gradient buffer (RCCL over xGMI on GPUs, issued in the step's own stream;
the same code runs over gloo on CPU tensors in the tests).  The 1/world factor is folded into the AdamW kernel's grad_scale.

Semantics follow the reference's accelerate/DDP setup (train.py:218-221): ``split_batches=True``
gives rank r the r-th contiguous slice of each global batch, every rank normalises its loss by its
local count of non-empty samples, gradients are averaged, parameters are broadcast from rank 0 once.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def init_from_env():
    """What ``Accelerator()`` does for the reference (train.py:218-221) when the process was started by ``torchrun`` /
    ``accelerate launch``: join the process group named by RANK / WORLD_SIZE / MASTER_* and select this rank's GPU.
    The group is the CONTROL plane only (gloo: barriers, the RCCL unique id, the fallback exchange); gradients travel
    through the engine's own RCCL communicator. Returns (rank, world, device string or None when no launcher is
    present). Idempotent."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not (dist.is_available() and dist.is_initialized()):
        return 0, 1, None
    if not dist.is_initialized():
        dist.init_process_group("gloo")
    rank = dist.get_rank()
    device = None
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        device = f"cuda:{local}"
    return rank, dist.get_world_size(), device


def shard_batch(batch, rank, world, pad=False, batch_size=None, first_batch=None):
    """accelerate BatchSamplerShard(split_batches=True) (ACC:data_loader.py, the reference's train.py:220): contiguous equal
    slices of the collated batch. ``batch`` = (labels [B,S], masked [B,S], lengths list, masked_indices list).
    ``pad``: what happens to a SHORT batch (the last one of a ``drop_last=False`` loader, i.e. validation):
      * with ``batch_size`` and ``first_batch`` (the pass's first collated batch, same 4-tuple): accelerate's
        ``even_batches=True`` — the short batch is completed to the full ``batch_size`` with the samples of the pass's
        first batch, in order (cycled if needed), and every rank takes ``batch_size // world`` of it. Rows of the two
        batches may have different padded widths (each is padded to its own longest sample): the narrower is zero-padded;
      * without them: completed only up to the next multiple of ``world`` with its OWN first samples — a deliberate
        simplification for callers that have no pass context (per-rank sample count and content of that one batch then
        differ from accelerate's; the training batches, which are always full, are unaffected)."""
    labels, masked, lengths, idx = batch
    B = len(lengths)
    if pad and batch_size is not None and first_batch is not None and B < batch_size:
        fl, fm, flen, fidx = first_batch
        nf = len(flen)
        take = [i % nf for i in range(batch_size - B)]
        S = max(labels.shape[1], fl.shape[1])
        widen = lambda a: np.pad(a, ((0, 0), (0, S - a.shape[1])))
        labels = np.concatenate([widen(np.asarray(labels)), widen(np.asarray(fl))[take]])
        masked = np.concatenate([widen(np.asarray(masked)), widen(np.asarray(fm))[take]])
        lengths = list(lengths) + [flen[i] for i in take]
        idx = list(idx) + [fidx[i] for i in take]
        B = batch_size
    elif B % world and pad:
        extra = [i % B for i in range(world - B % world)]
        take = list(range(B)) + extra
        labels, masked = labels[take], masked[take]
        lengths, idx = [lengths[i] for i in take], [idx[i] for i in take]
        B = len(take)
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by the world size {world}")
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    return labels[sl], masked[sl], list(lengths[sl]), list(idx[sl])


class GradReducer:
    """Sum all-reduce of a flat gradient tensor through ``torch.distributed`` — the FALLBACK exchange, used when the
    engine's own RCCL communicator cannot be created (ranks sharing one device on the one-GPU test box: gloo) and by
    the CPU world-2 tests. On the CURRENT stream by default, or in contiguous pieces on a side stream
    (``side_stream=True``).

    The product exchange is NOT this class: ``plb_loss_fwd_bwd`` issues the all-reduce itself, piecewise, on the
    engine's communication stream, each piece behind the weight-gradient GEMM that completed it
    (``csrc/engine.cpp: reduce_piece``); ``plb_allreduce_grads`` joins it. This class reduces the whole
    buffer after the backward because a host-side collective cannot be ordered between the launches of one C call."""

    def __init__(self, group=None, device=None, force=False, side_stream=False):
        self.group = group
        self.rank, self.world = world_info(group)
        # force: issue the collectives even at world size 1 (rehearsal of the N > 1 path on one GPU)
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self.stream = None
        if side_stream and self.active and device is not None and torch.device(device).type == "cuda":
            self.stream = torch.cuda.Stream(device=device)

    def broadcast_(self, flat, src=0):
        if self.active:
            dist.broadcast(flat, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)

    def all_reduce_(self, flat, pieces=None):
        """flat: 1-D tensor; pieces: optional list of (begin, end) element ranges reduced as separate
        collectives (default: the whole tensor in one)."""
        if not self.active:
            return
        pieces = pieces or [(0, flat.numel())]
        if self.stream is None:
            for a, b in pieces:
                dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
            return
        main = torch.cuda.current_stream(flat.device)
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            for a, b in pieces:
                dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
        main.wait_stream(self.stream)
