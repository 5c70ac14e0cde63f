"""Data-parallel plumbing of the hot path: one process per GPU, replicated
parameters, the global batch split by sample across ranks and ONE exchange step per training step —
a sum all-reduce of the flat gradient buffer (RCCL over xGMI on GPUs, issued in the step's own stream;
the same code runs over gloo on CPU tensors in the tests).  The 1/world factor is folded into the AdamW kernel's grad_scale.

Semantics follow the reference's accelerate/DDP setup (train.py:218-221): ``split_batches=True``
gives rank r the r-th contiguous slice of each global batch, every rank normalises its loss by its
local count of non-empty samples, gradients are averaged, parameters are broadcast from rank 0 once.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_batch(batch, rank, world):
    """accelerate BatchSamplerShard(split_batches=True): contiguous equal slices of the collated batch.
    ``batch`` = (labels [B,S], masked [B,S], lengths list, masked_indices list)."""
    labels, masked, lengths, idx = batch
    B = len(lengths)
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by the world size {world}")
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    return labels[sl], masked[sl], list(lengths[sl]), list(idx[sl])


class GradReducer:
    """Sum all-reduce of a flat gradient tensor on the CURRENT stream (default), or in contiguous pieces on a
    side stream (``side_stream=True``) for callers that have later main-stream work to overlap.

    The training step uses the default. Its batched weight-gradient GEMMs make every large gradient final only
    at the end of the backward, AdamW needs the reduced gradients right after, so there is
    nothing for a side stream to overlap — and on MI355X the two cross-stream waits of a side-stream
    collective cost 0.1–0.5 ms per step (tools/dist_overhead.py, world size 1), against 0 for the
    in-stream call. Splitting the 23 MB into pieces ordered by completion was priced too: the gradients that
    finish early are small (head 0.6 MB, embeddings 0.8 MB) and the flat order interleaves late and early
    tensors, so it trades one collective for 7–8 latency-bound ones."""

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
