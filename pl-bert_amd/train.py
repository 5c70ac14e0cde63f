"""Training-step driver of the hot path: the loop body of the reference (train.py:350-357) as three
device calls — loss+backward (plb_loss_fwd_bwd), gradient all-reduce (RCCL via torch.distributed,
only when world_size > 1), AdamW (plb_adamw_step).

Data parallelism follows the reference's DDP semantics: one process per GPU,
replicated parameters and optimizer state, each rank normalises its loss by its LOCAL count of
non-empty samples (train.py:129) and gradients are averaged over ranks; the logged loss is the
local one (train.py:395-410).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from .data import masked_indices_to_csr
from .engine import HipEngine
from .init import reference_init_state_dict


@dataclass
class StagedBatch:
    """One collated batch resident on the device (what dataloader.py:297 returns, staged once)."""
    masked: torch.Tensor          # int64 [B,S]
    labels: torch.Tensor          # int64 [B,S]
    lengths: torch.Tensor | None  # int32 [B] or None when nothing is padded
    offsets: torch.Tensor         # int32 [B+1]
    flat: torch.Tensor            # int32 [n_masked]
    n_masked: int
    n_tokens: int


def validate_batch(labels, masked, lengths, masked_indices, vocab_size):
    """Host checks the kernels rely on (they index without bounds checks)."""
    labels = np.asarray(labels)
    masked = np.asarray(masked)
    if labels.shape != masked.shape or labels.ndim != 2:
        raise ValueError("labels / masked must both be [B,S]")
    B, S = masked.shape
    if len(lengths) != B or len(masked_indices) != B:
        raise ValueError("lengths / masked_indices must have one entry per sample")
    if masked.min(initial=0) < 0 or masked.max(initial=0) >= vocab_size or labels.min(initial=0) < 0 or \
            labels.max(initial=0) >= vocab_size:
        raise ValueError("phoneme id outside the vocabulary")
    for b, (L, idx) in enumerate(zip(lengths, masked_indices)):
        if not (1 <= int(L) <= S):
            raise ValueError(f"sample {b}: length {L} outside [1, {S}]")
        if len(idx):
            a = np.asarray(idx)
            if a.min() < 0 or a.max() >= int(L):
                raise ValueError(f"sample {b}: masked index outside [0, length)")
            if len(np.unique(a)) != len(a):
                raise ValueError(f"sample {b}: duplicate masked index")


class PLBertTrainer:
    """PhonemeOnlyModel + AdamW(lr) of train.py:266-272 on one GPU, optionally data parallel."""

    def __init__(self, cfg, num_phonemes, max_batch=32, max_seq=512, lr=7e-5, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.01, device=None, seed=0, state_dict=None, process_group=None):
        self.engine = HipEngine(cfg, num_phonemes, 0, max_batch=max_batch, max_seq=max_seq, device=device)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        sd = state_dict if state_dict is not None else reference_init_state_dict(cfg, num_phonemes, 0, seed=seed)
        self.engine.load_state_dict(sd)
        if self.world > 1:  # DDP's start-up broadcast of rank 0's parameters
            dist.broadcast(self.engine.params, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                           group=process_group)
            self.engine.sync_weights()
        self._comm_stream = torch.cuda.Stream(device=self.engine.device) if self.world > 1 else None

    def stage_batch(self, labels, masked, lengths, masked_indices, validate=True):
        if validate:
            validate_batch(labels, masked, lengths, masked_indices, self.engine.cfg.vocab_size)
        dev = self.engine.device
        masked_t = torch.as_tensor(np.asarray(masked), dtype=torch.int64).to(dev)
        labels_t = torch.as_tensor(np.asarray(labels), dtype=torch.int64).to(dev)
        B, S = masked_t.shape
        lens = np.asarray(lengths, dtype=np.int32)
        lengths_t = None if (lens == S).all() else torch.from_numpy(lens).to(dev)
        off, flat = masked_indices_to_csr(masked_indices)
        return StagedBatch(masked_t, labels_t, lengths_t, torch.from_numpy(off).to(dev), torch.from_numpy(flat).to(dev),
                           int(off[-1]), int(lens.sum()))

    def loss_and_grads(self, batch: StagedBatch):
        return self.engine.loss_fwd_bwd(batch.masked, batch.labels, batch.lengths, batch.offsets, batch.flat,
                                        batch.n_masked)

    def all_reduce_grads(self):
        """Sum the trainable gradient range over ranks on a side stream (RCCL over xGMI); the AdamW
        kernel applies the 1/world factor."""
        if self.world == 1:
            return
        main = torch.cuda.current_stream(self.engine.device)
        self._comm_stream.wait_stream(main)
        with torch.cuda.stream(self._comm_stream):
            dist.all_reduce(self.engine.grads[: self.engine.trainable], op=dist.ReduceOp.SUM, group=self.group)
        main.wait_stream(self._comm_stream)

    def step(self, batch: StagedBatch):
        """zero_grad + backward + optimizer.step of train.py:355-357; returns the local loss (device)."""
        loss = self.loss_and_grads(batch)
        if batch.n_masked == 0 and self.world == 1:
            return loss  # reference: zero-loss fallback has no graph, the optimizer sees no gradients
        self.all_reduce_grads()
        self.step_count += 1
        self.engine.adamw_step(self.step_count, self.lr, self.betas, self.eps, self.weight_decay,
                               grad_scale=1.0 / self.world)
        return loss
