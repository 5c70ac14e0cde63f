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

import os

import numpy as np
import torch

from .data import masked_indices_to_csr
from .dist import GradReducer
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
    token_ids: torch.Tensor | None = None  # int64 [B,S]: grapheme-token targets of the 4-tuple Collater (dual-head)


def validate_token_ids(token_ids, shape, lengths, num_tokens):
    """Targets of the token head must be class ids at every valid position (the kernel indexes with them)."""
    tok = np.asarray(token_ids)
    if tok.shape != tuple(shape):
        raise ValueError("token_ids must have the shape of the phoneme batch")
    valid = np.arange(shape[1])[None, :] < np.asarray(lengths)[:, None]
    if tok[valid].min(initial=0) < 0 or tok[valid].max(initial=0) >= num_tokens:
        raise ValueError("token id outside [0, num_tokens)")


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


def _rewind_steps(owner):
    """HandoffTimeout callback for an object that counts optimizer steps (``step_count``): the device skipped
    ``err.skipped_updates`` of them. Holds the owner weakly (the engine must not keep its trainer alive)."""
    import weakref
    ref = weakref.ref(owner)

    def cb(err):
        o = ref()
        if o is not None:
            o.step_count = max(0, o.step_count - err.skipped_updates)
    return cb


class PLBertTrainer:
    """PhonemeOnlyModel + AdamW(lr) of train.py:266-272 on one GPU, optionally data parallel.
    ``num_tokens > 0`` builds MultiTaskModel's second head (model.py:11); batches staged with ``token_ids``
    then train both heads (loss = phoneme loss + token loss, see include/plbert.h plb_loss_fwd_bwd_dual)."""

    def __init__(self, cfg, num_phonemes, max_batch=32, max_seq=512, lr=7e-5, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.01, device=None, seed=0, state_dict=None, process_group=None, force_collectives=False,
                 num_tokens=0, comm="auto", overlap=True):
        """``comm``: how the gradient exchange of a data-parallel run travels.
        "rccl"  — the engine's own RCCL communicator behind the C ABI (plb_comm_init / plb_allreduce_grads): the
                  all-reduce is issued by plb_loss_fwd_bwd itself, piece by piece on the engine's communication
                  stream while the remaining weight-gradient GEMMs run (``overlap``); torch.distributed only
                  carries the 128-byte unique id. Needs one GPU per rank.
        "torch" — ``torch.distributed.all_reduce`` on the flat buffer (dist.GradReducer): gloo in the CPU / shared-GPU
                  tests, or any backend the caller initialised.
        "auto"  — "rccl" when the process group spans more than one rank (or ``force_collectives``) and every rank of
                  this node has a GPU of its own, else "torch"."""
        self.engine = HipEngine(cfg, num_phonemes, num_tokens, max_batch=max_batch, max_seq=max_seq, device=device)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.reducer = GradReducer(process_group, device=self.engine.device, force=force_collectives)
        self.world = self.reducer.world
        sd = state_dict if state_dict is not None else reference_init_state_dict(cfg, num_phonemes, num_tokens, seed=seed)
        self.engine.load_state_dict(sd)
        # a step whose fused LayerNorm hand-off timed out is skipped on the device (include/plbert.h: plb_status) and
        # raises HandoffTimeout from the next engine call: the bias-correction count goes back by the skipped updates
        self.engine._on_handoff_timeout.append(_rewind_steps(self))
        self.comm = "none"
        if self.reducer.active:
            if comm == "auto":   # PLBERT_COMM=rccl|torch overrides (tests: the engine's own exchange through a stand-in library)
                comm = os.environ.get("PLBERT_COMM") or ("rccl" if own_gpu_per_rank() else "torch")
            if comm == "rccl":
                uid = exchange_unique_id(HipEngine.comm_unique_id if self.reducer.rank == 0 else None, process_group)
                self.engine.comm_init(uid, self.reducer.rank, self.world)
                self.engine.set_grad_overlap(overlap)
                self.engine.broadcast_params(0)  # DDP's start-up broadcast of rank 0's parameters
            elif comm == "torch":
                self.reducer.broadcast_(self.engine.params)
                self.engine.sync_weights()
            else:
                raise ValueError(f"comm must be 'auto', 'rccl' or 'torch', not {comm!r}")
            self.comm = comm

    def stage_batch(self, labels, masked, lengths, masked_indices, validate=True, token_ids=None):
        if validate:
            validate_batch(labels, masked, lengths, masked_indices, self.engine.cfg.vocab_size)
            if token_ids is not None:
                validate_token_ids(token_ids, np.asarray(masked).shape, lengths, self.engine.num_tokens)
        dev = self.engine.device
        masked_t = torch.as_tensor(np.asarray(masked), dtype=torch.int64).to(dev)
        labels_t = torch.as_tensor(np.asarray(labels), dtype=torch.int64).to(dev)
        B, S = masked_t.shape
        lens = np.asarray(lengths, dtype=np.int32)
        lengths_t = None if (lens == S).all() else torch.from_numpy(lens).to(dev)
        off, flat = masked_indices_to_csr(masked_indices)
        tok_t = None if token_ids is None else torch.as_tensor(np.asarray(token_ids), dtype=torch.int64).to(dev)
        return StagedBatch(masked_t, labels_t, lengths_t, torch.from_numpy(off).to(dev), torch.from_numpy(flat).to(dev),
                           int(off[-1]), int(lens.sum()), tok_t)

    def loss_and_grads(self, batch: StagedBatch):
        return self.engine.loss_fwd_bwd(batch.masked, batch.labels, batch.lengths, batch.offsets, batch.flat,
                                        batch.n_masked, token_ids=batch.token_ids)

    def all_reduce_grads(self, dual=False):
        """Sum the trainable gradient range over ranks (RCCL over xGMI, one collective in the step's stream:
        see dist.GradReducer); the AdamW kernel applies the 1/world factor. A dual-head step also carries
        the token head's gradients."""
        if self.comm == "rccl":
            self.engine.allreduce_grads()  # joins the pieces plb_loss_fwd_bwd issued (or reduces now: overlap off)
            return
        self.reducer.all_reduce_(self.engine.grads[: self.engine.trainable])
        if dual:
            a, b = self.engine.token_range
            self.reducer.all_reduce_(self.engine.grads[a:b])
        if self.reducer.active:  # the health word travels with the gradients: every rank skips a poisoned update, or none
            self.engine.status_exchange(self.reducer.all_reduce_)

    def step(self, batch: StagedBatch):
        """zero_grad + backward + optimizer.step of train.py:355-357; returns the local loss (device).
        HandoffTimeout (never observed; engine.py) is raised by the engine call that follows the failed step's completion
        on the device: the failed step's update has been left out on EVERY rank (the health word is agreed between the
        ranks inside the step) and all replicas are bit-identical, so the caller may run the batch again. A loop that
        reads each loss back (run.train_loop) sees it exactly at the failed step, on every rank at the same step; a
        loop that does not may see it one step late and — in a data-parallel run — at different steps on different
        ranks: such a caller must treat it as the end of the run (resume from the replicas, which are consistent)."""
        loss = self.loss_and_grads(batch)
        dual = batch.token_ids is not None
        if batch.n_masked == 0 and self.world == 1 and not dual:
            return loss  # reference: zero-loss fallback has no graph, the optimizer sees no gradients
        self.all_reduce_grads(dual)
        self.step_count += 1
        self.engine.adamw_step(self.step_count, self.lr, self.betas, self.eps, self.weight_decay,
                               grad_scale=1.0 / self.world)
        return loss


def own_gpu_per_rank():
    """True when every rank launched on this node can have a GPU to itself (RCCL refuses two ranks on one device)."""
    import os
    import torch.distributed as dist
    local_world = os.environ.get("LOCAL_WORLD_SIZE")
    if local_world is None:  # not under a launcher: assume every rank of the group lives on this node
        local_world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return torch.cuda.device_count() >= int(local_world)


_uid_round = [0]


def exchange_unique_id(make, group=None):
    """Rank 0 of ``group`` creates the RCCL unique id (``make()``), everyone receives its 128 bytes. Over the
    process group's key-value store when there is one (no device traffic, so torch creates no communicator of its
    own), else as a broadcast object."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    store = None
    if group is None:
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
    if store is not None:
        key = f"plbert_rccl_uid_{_uid_round[0]}"
        _uid_round[0] += 1
        if rank == 0:
            store.set(key, make())
        return bytes(store.get(key))
    box = [make() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return bytes(box[0])


# ---- drop-in forms of the reference's step functions ---------------------------------------------------
class _FusedLoss(torch.autograd.Function):
    """Bridges the fused fwd+bwd call into autograd so the reference loop body
    ``loss = process_batch(...); optimizer.zero_grad(); accelerator.backward(loss); optimizer.step()``
    (train.py:352-357) runs unchanged: forward computes loss AND gradients in one engine call,
    backward hands each Parameter its slice of the engine's flat gradient buffer."""

    @staticmethod
    def forward(ctx, engine, names, batch, *params):
        loss = engine.loss_fwd_bwd(batch.masked, batch.labels, batch.lengths, batch.offsets, batch.flat, batch.n_masked,
                                   token_ids=batch.token_ids)
        ctx.engine, ctx.names, ctx.dual = engine, names, batch.token_ids is not None
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        ta, tb = eng.token_range
        grads = []
        for n in ctx.names:
            off, size, shp = eng.layout[n]
            has = off + size <= eng.trainable or (ctx.dual and off >= ta)  # pooler: never; token head: dual steps
            grads.append(eng.grads[off:off + size].view(shp) if has else None)
        # d(loss)/d(param) is already in the flat buffer; scale by the upstream gradient on the device
        # (1.0 for loss.backward(); no host sync to find out)
        eng.grads[: eng.trainable].mul_(grad_out)
        if ctx.dual:
            eng.grads[ta:tb].mul_(grad_out)
        return (None, None, None, *grads)


def stage_reference_batch(engine, batch, validate=True):
    """(labels, masked, lengths, masked_indices) as PhonemeOnlyCollater returns it, or the 5-tuple
    (token_ids, labels, masked, lengths, masked_indices) of Collater (dataloader.py:200-223) -> StagedBatch."""
    token_ids = None
    if len(batch) == 5:
        token_ids, *batch = batch
    labels, masked, lengths, idx = batch
    if validate:
        validate_batch(labels, masked, lengths, idx, engine.cfg.vocab_size)
        if token_ids is not None:
            validate_token_ids(token_ids, np.asarray(masked).shape, lengths, engine.num_tokens)
    dev = engine.device
    masked_t = torch.as_tensor(np.asarray(masked), dtype=torch.int64).to(dev)
    labels_t = torch.as_tensor(np.asarray(labels), dtype=torch.int64).to(dev)
    S = masked_t.shape[1]
    lens = np.asarray(lengths, dtype=np.int32)
    lengths_t = None if (lens == S).all() else torch.from_numpy(lens).to(dev)
    off, flat = masked_indices_to_csr(idx)
    tok_t = None if token_ids is None else torch.as_tensor(np.asarray(token_ids), dtype=torch.int64).to(dev)
    return StagedBatch(masked_t, labels_t, lengths_t, torch.from_numpy(off).to(dev), torch.from_numpy(flat).to(dev),
                       int(off[-1]), int(lens.sum()), tok_t)


def device_mask_batch(labels, lengths=None, seed=1, step=0, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1,
                      device=None):
    """Fast mode of the masking path: word-level mask / replace / keep ON THE DEVICE
    (plb_mask_batch). Same decision tree and probabilities as MaskedPhonemeDataset (dataloader.py:83-108),
    Philox randomness keyed by (seed, step, sample, word): distribution-matched, not the reference's bit
    stream. ``labels`` int64 [B,S] = unmasked phoneme ids with the separator 186 after each word, zero
    padded; returns a StagedBatch (one small device->host read for the masked count)."""
    import ctypes as C

    from . import _lib
    from .symbols import MASK_ID, SEPARATOR_ID

    L = _lib.lib()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    labels_t = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(dev, torch.int64).contiguous()
    B, S = labels_t.shape
    lengths_t = None
    if lengths is not None:
        lens = np.asarray(lengths, dtype=np.int32)
        lengths_t = None if (lens == S).all() else torch.from_numpy(lens).to(dev)
    masked = torch.empty_like(labels_t)
    offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
    flat = torch.empty(B * S, dtype=torch.int32, device=dev)
    scratch = torch.empty(B + B * S, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(L.plb_mask_batch(labels_t.data_ptr(), None if lengths_t is None else lengths_t.data_ptr(), B, S,
                                int(seed), int(step), word_pred_prob, phoneme_mask_prob, replace_prob, MASK_ID,
                                SEPARATOR_ID, masked.data_ptr(), offsets.data_ptr(), flat.data_ptr(), scratch.data_ptr(),
                                stream), "plb_mask_batch")
    n = int(offsets[B].item())
    n_tokens = int(B * S if lengths is None else np.asarray(lengths).sum())
    return StagedBatch(masked, labels_t, lengths_t, offsets, flat[:n], n, n_tokens)


def device_apply_mask(records, device=None, word_separator=None):
    """Bit-exact masking through the GPU: ``records`` are ``MaskedPhonemeDataset.decisions(i)``
    dicts — the reference's random draws, made on the host in the reference's order — and plb_apply_mask does the
    integer work (crop, mask / replace, separator handling, index re-basing, zero padding, length-descending batch
    order) on the device. Returns (StagedBatch, labels, masked, lengths, token_ids|None): identical to what
    ``PhonemeOnlyCollater()([ds[i] ...])`` / ``Collater()`` return (tests/test_gpu_apply_mask.py)."""
    import ctypes as C

    from . import _lib
    from .data import collate_decisions
    from .symbols import MASK_ID

    L = _lib.lib()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    c = collate_decisions(records)
    B, S = c["B"], c["S"]
    if B < 1 or S < 1:
        raise ValueError("device_apply_mask needs at least one non-empty sample")
    with torch.cuda.device(dev):
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
        ids, repl = up(c["ids"]), up(c["repl"])
        sample_off, word_off = up(c["sample_off"]), up(c["word_off"])
        word_begin, word_len, action = up(c["word_begin"]), up(c["word_len"]), up(c["action"])
        crop = up(c["crop_start"])
        wtok = None if c["word_token"] is None else up(c["word_token"])
        labels = torch.empty((B, S), dtype=torch.int64, device=dev)
        masked = torch.empty_like(labels)
        tokens = torch.empty_like(labels) if wtok is not None else None
        lens = torch.empty(B, dtype=torch.int32, device=dev)
        offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
        flat = torch.empty(B * S, dtype=torch.int32, device=dev)
        scratch = torch.empty(B + B * S, dtype=torch.int32, device=dev)
        p = lambda t: None if t is None else t.data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.plb_apply_mask(ids.data_ptr(), sample_off.data_ptr(), word_off.data_ptr(), word_begin.data_ptr(),
                                    word_len.data_ptr(), action.data_ptr(), repl.data_ptr(), p(wtok),
                                    int(word_separator if word_separator is not None else 0), crop.data_ptr(), B, S,
                                    MASK_ID, labels.data_ptr(), masked.data_ptr(), p(tokens), lens.data_ptr(),
                                    offsets.data_ptr(), flat.data_ptr(), scratch.data_ptr(), stream), "plb_apply_mask")
        n = int(offsets[B].item())
    lengths = c["lengths"]
    lengths_t = None if all(l == S for l in lengths) else lens
    return StagedBatch(masked, labels, lengths_t, offsets, flat[:n], n, int(sum(lengths)), tokens), labels, masked, lengths, tokens


def process_batch(model, batch, criterion=None, accelerator=None):
    """train.py:381-390 — ``batch = (phoneme_labels, masked_phonemes, input_lengths, masked_indices)``.
    ``criterion`` / ``accelerator`` are accepted for signature compatibility: the loss is the
    reference's calculate_phoneme_loss with nn.CrossEntropyLoss() (train.py:107-131, 215), computed by
    the HIP engine. Returns a 0-dim tensor; under autograd its ``backward()`` fills ``param.grad``.
    A 5-tuple batch (Collater, dataloader.py:200-223: token_ids first) on a MultiTaskModel trains both heads:
    phoneme loss + token loss (plb_loss_fwd_bwd_dual)."""
    engine = model.engine
    staged = batch if isinstance(batch, StagedBatch) else stage_reference_batch(engine, batch)
    if staged.n_masked == 0 and staged.token_ids is None:  # train.py:129
        return torch.tensor(0.0, device=engine.device, requires_grad=True)
    names, params = [], []
    for n, p in model.named_parameters():
        names.append(n)
        params.append(p)
    if torch.is_grad_enabled():
        return _FusedLoss.apply(engine, names, staged, *params)
    # validate() (train.py:288-304): forward + loss only; the gradient buffer is left alone
    return engine.loss_fwd(staged.masked, staged.labels, staged.lengths, staged.offsets, staged.flat,
                           staged.n_masked, token_ids=staged.token_ids)[0].clone()


class AdamW:
    """``torch.optim.AdamW(model.parameters(), lr=...)`` of train.py:272 as one fused HIP launch over the
    model's flat buffers. ``params`` must be the parameters of ONE plbert_amd model (it identifies the
    engine); state_dict()/load_state_dict() carry step + both moments (train.py:417-421 'optimizer')."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, model=None):
        self.param_list = list(params)
        if model is None:
            raise ValueError("pass model=<PhonemeOnlyModel|MultiTaskModel>: the fused optimizer updates the model's "
                             "flat parameter buffer")
        self.engine = model.engine
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.step_count = 0
        self.grad_scale = 1.0
        self.engine._on_handoff_timeout.append(_rewind_steps(self))
        by_id = {id(p): n for n, p in model.named_parameters()}
        self._names = [by_id[id(p)] for p in self.param_list]

    def zero_grad(self, set_to_none=True):
        for p in self.param_list:
            p.grad = None

    def step(self):
        # parameters without a gradient are skipped as torch does: a step with no gradient at all (the
        # zero-loss fallback) changes nothing; otherwise the trainable range is updated in one launch
        if all(p.grad is None for p in self.param_list):
            return
        e = self.engine
        for n, p in zip(self._names, self.param_list):
            # autograd may have handed the Parameter a copy of its gradient slice (or user code replaced
            # / clipped .grad): whatever .grad holds now is what the fused update must consume
            off, size, shp = e.layout[n]
            in_range = off + size <= e.trainable or off >= e.token_range[0]
            if p.grad is not None and in_range and p.grad.data_ptr() != e.grads.data_ptr() + 4 * off:
                e.grads[off:off + size].view(shp).copy_(p.grad)
        self.step_count += 1
        d = self.defaults
        self.engine.adamw_step(self.step_count, d["lr"], d["betas"], d["eps"], d["weight_decay"], self.grad_scale)

    def state_dict(self):
        """torch.optim.AdamW layout ({'state': {index: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}) with
        parameter indices in ``model.parameters()`` order, so the 'optimizer' entry of a checkpoint
        (train.py:417-421) can be loaded by either implementation. Parameters that never received a
        gradient (the pooler; the token head before its first dual-head step) have no state, as in torch; the token
        head carries its own step count (it may have started later than the encoder)."""
        e = self.engine
        d = self.defaults
        state = {}
        ta = e.token_range[0]
        tok_steps = e.token_head_steps
        for i, n in enumerate(self._names):
            off, size, shp = e.layout[n]
            if off + size <= e.trainable:
                steps = self.step_count
            elif off >= ta and e.num_tokens:
                steps = tok_steps
            else:
                steps = 0
            if steps > 0:
                state[i] = {"step": torch.tensor(float(steps)),
                            "exp_avg": e.exp_avg[off:off + size].view(shp).clone(),
                            "exp_avg_sq": e.exp_avg_sq[off:off + size].view(shp).clone()}
        group = {"lr": d["lr"], "betas": tuple(d["betas"]), "eps": d["eps"], "weight_decay": d["weight_decay"],
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": True, "params": list(range(len(self.param_list)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        e = self.engine
        e._bind()
        if "param_groups" in sd:
            g = sd["param_groups"][0]
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in g:
                    self.defaults[k] = tuple(g[k]) if k == "betas" else g[k]
            steps, tok_steps = set(), set()
            e.exp_avg.zero_()
            e.exp_avg_sq.zero_()
            ta = e.token_range[0]
            for i, st in sd["state"].items():
                off, size, shp = e.layout[self._names[int(i)]]
                is_tok = bool(e.num_tokens) and off >= ta
                if off + size > e.trainable and not is_tok:
                    continue  # the pooler never trains
                e.exp_avg[off:off + size].view(shp).copy_(st["exp_avg"])
                e.exp_avg_sq[off:off + size].view(shp).copy_(st["exp_avg_sq"])
                (tok_steps if is_tok else steps).add(int(float(st["step"])))
            if len(steps) > 1 or len(tok_steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)} / token head {sorted(tok_steps)}): "
                                 "the fused AdamW keeps one step for the encoder + phoneme head and one for the token head")
            self.step_count = steps.pop() if steps else 0
            e.token_head_steps = tok_steps.pop() if tok_steps else 0
            return
        self.step_count = int(sd["step"])  # compact form written by early versions of this class
        e.exp_avg[: e.trainable].copy_(sd["exp_avg"])
        e.exp_avg_sq[: e.trainable].copy_(sd["exp_avg_sq"])
