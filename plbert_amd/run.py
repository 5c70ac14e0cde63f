"""Run management of the hot path's caller: what ``train.train(args)`` does around the step in the
reference — run directory + resume-by-directory (train.py:174-210), latest-checkpoint discovery (train.py:46-79),
validation before the first step and at every ``save_interval`` (train.py:288-336,344,369-373), ``step_N.pth`` files
(train.py:412-425) — around the MI355X-native step (``PLBertTrainer``: plb_loss_fwd_bwd -> RCCL exchange -> plb_adamw_step)
and input pipeline (``DeviceFeeder``).

Differences from the reference, on purpose: metrics go to ``<run dir>/metrics.jsonl`` and stdout instead of wandb
(out of scope); the dataset may be handed in as any sequence of ``{'phonemes': [...]}`` rows
(the reference's hub download, train.py:245, needs a network); ``num_workers`` and device-side masking are options of
``training_params`` (``num_workers``, ``device_masking``) that default to the reference's behaviour (0 / off).

    python -m plbert_amd.run --config_path configs/config.yml --run_name default        # train.py:27-32
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
from collections import deque

import numpy as np
import torch
import torch.distributed as dist
import yaml

from .checkpoint import find_latest_checkpoint, optimizer_state_from_flat, optimizer_state_to_flat
from .config import albert_config_from_yaml
from .data import build_dataloader
from .dist import init_from_env, shard_batch, world_info
from .symbols import symbols

MAX_EPOCHS = 10  # train.py:145


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", type=str, default="configs/config.yml")
    ap.add_argument("--run_name", type=str, default="default")
    return vars(ap.parse_args(argv))


def setup_config_and_directories(args, config_path):
    """Run folder decides: an existing folder that holds a copy of the config RESUMES with that copy; an existing folder
    without one is cleaned of ``step_*`` files and starts fresh; otherwise the folder is created (train.py:174-210).
    Returns (config, log_dir, resuming). With several ranks only rank 0 touches the folder (the reference guards the
    same block with ``accelerator.is_main_process``, train.py:181); the others read its decision after a barrier."""
    rank, world = world_info()
    if world > 1:
        decided = [None]
        if rank == 0:
            decided[0] = _setup_run_dir(args, config_path)
        dist.broadcast_object_list(decided, src=0)
        return decided[0]
    return _setup_run_dir(args, config_path)


def _setup_run_dir(args, config_path):
    with open(config_path) as f:
        given = yaml.safe_load(f)
    log_dir = os.path.join(given["training_params"]["output_dir"], args["run_name"])
    kept = os.path.join(log_dir, os.path.basename(config_path))
    if os.path.isdir(log_dir) and os.path.exists(kept):
        with open(kept) as f:
            return yaml.safe_load(f), log_dir, True
    if os.path.isdir(log_dir):
        for name in os.listdir(log_dir):
            if name.startswith("step_"):
                os.remove(os.path.join(log_dir, name))
    os.makedirs(log_dir, exist_ok=True)
    shutil.copy(config_path, kept)
    return given, log_dir, False


class _Log:
    """Rank-0 metrics sink: one JSON object per event, to stdout and ``metrics.jsonl`` (the reference logs the same
    keys to wandb: phoneme_loss, phoneme_loss_avg, val_phoneme_loss, epoch, step — train.py:326-331,398-410)."""

    def __init__(self, log_dir, main):
        self.main = main
        self.f = open(os.path.join(log_dir, "metrics.jsonl"), "a") if main else None

    def __call__(self, **kv):
        if self.main:
            line = json.dumps(kv)
            print(line, flush=True)
            self.f.write(line + "\n")
            self.f.flush()


def _state_for_file(trainer, step, epoch):
    """``{'net','step','epoch','optimizer'}`` with the optimizer entry in torch.optim.AdamW's layout, as train.py:416-421
    writes it: state indices count the reference's ``model.parameters()`` order (checkpoint.REFERENCE_PARAM_ORDER), so
    the reference's ``optimizer.load_state_dict`` binds every moment to the tensor it belongs to."""
    eng = trainer.engine
    tok0 = eng.token_range[0]

    def steps_of(name, off, size):
        if off + size <= eng.trainable:
            return trainer.step_count
        return eng.token_head_steps if eng.num_tokens and off >= tok0 else 0

    state, n = optimizer_state_from_flat(eng.layout, eng.exp_avg, eng.exp_avg_sq, steps_of)
    group = {"lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.weight_decay,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "decoupled_weight_decay": True, "params": list(range(n))}
    return {"net": {k: v.cpu() for k, v in eng.state_dict().items()}, "step": step, "epoch": epoch,
            "optimizer": {"state": state, "param_groups": [group]}}


def save_checkpoint(trainer, current_step, log_dir, current_epoch, main=True):
    path = os.path.join(log_dir, f"step_{current_step}.pth")
    if main:
        torch.save(_state_for_file(trainer, current_step, current_epoch), path)
        print(f"Checkpoint saved at: {path}", flush=True)
    return path


def load_checkpoint(trainer, path):
    """Weights (``module.`` prefixes stripped, non-strict: train.py:98-100) and, when present, the AdamW state — of a
    file this driver wrote or of a reference ``step_N.pth`` (indices in ``model.parameters()`` order)."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    eng = trainer.engine
    eng.load_state_dict({k.replace("module.", ""): v for k, v in ck["net"].items()}, strict=False)
    opt = ck.get("optimizer")
    if opt and opt.get("state"):
        tok0 = eng.token_range[0]

        def is_tok(off):
            return bool(eng.num_tokens) and off >= tok0

        steps = optimizer_state_to_flat(opt["state"], eng.layout, eng.exp_avg, eng.exp_avg_sq,
                                        lambda n, off, size: off + size <= eng.trainable or is_tok(off))  # not the pooler
        main_steps = {v for n, v in steps.items() if not is_tok(eng.layout[n][0])}
        tok_steps = {v for n, v in steps.items() if is_tok(eng.layout[n][0])}
        trainer.step_count = max(main_steps) if main_steps else 0
        eng.token_head_steps = max(tok_steps) if tok_steps else 0
    print(f"Checkpoint {path} loaded.", flush=True)
    return int(ck.get("step", 0))


def initialize_model(config, log_dir, resuming, device=None, max_batch=None):
    """train.py:261-286: model + AdamW(lr), optional ``pretrained_model``, then the run's latest checkpoint when
    resuming. Returns (trainer, current_step)."""
    from .train import PLBertTrainer
    cfg = albert_config_from_yaml(config, len(symbols))
    tp = config["training_params"]
    _, world = world_info()
    per_rank = max(1, int(tp["batch_size"]) // world)           # split_batches=True: batch_size is global (train.py:220)
    trainer = PLBertTrainer(cfg, num_phonemes=len(symbols), max_batch=max_batch or per_rank,
                            max_seq=int(config["dataset_params"]["max_seq_length"]), lr=float(tp["learning_rate"]),
                            device=device)
    if config["model_params"].get("pretrained_model"):
        print(f"Loading pretrained model from: {config['model_params']['pretrained_model']}")
        load_checkpoint(trainer, config["model_params"]["pretrained_model"])
    found, last = find_latest_checkpoint(log_dir)
    current_step = 0
    if found and resuming:
        current_step = load_checkpoint(trainer, os.path.join(log_dir, f"step_{last}.pth")) or last
    return trainer, current_step


def _feeder(loader, trainer, word_separator, budget=None):
    """ONE DeviceFeeder per (trainer, loader): its pinned buffers, device slots and copy stream are created once (a fresh
    set per validation pass was measured as tens of slow steps after every pass)."""
    import weakref

    from .pipeline import DeviceFeeder
    # keyed by the loader OBJECT (weakly): id() can be handed to a new loader once the old one has been collected
    cache = trainer.__dict__.setdefault("_feeders", weakref.WeakKeyDictionary())
    f = cache.get(loader)
    if f is None:
        f = cache[loader] = DeviceFeeder(loader, device=trainer.engine.device, vocab_size=trainer.engine.cfg.vocab_size,
                                         word_separator=word_separator, draw_budget=budget)
    return f


class _ShardedLoader:
    """This rank's contiguous slice of every collated batch of ``loader`` (accelerate's split_batches=True, train.py:220 —
    ``dist.shard_batch``; a short last validation batch is completed as accelerate's even_batches=True does). An iterable
    like the loader itself, so the same DeviceFeeder (pinned buffer, copy stream, double-buffered device slots) feeds a
    data-parallel rank: every rank still draws the masks of the WHOLE global batch from its own copy of the reference's
    streams, as the reference's ranks do."""

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world
        self.batch_size = getattr(loader, "batch_size", None)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        first = None
        for batch in self.loader:
            cur = (np.asarray(batch[0]), np.asarray(batch[1]), batch[2], batch[3])
            if first is None:
                first = cur
            yield shard_batch(cur, self.rank, self.world, pad=True, batch_size=self.batch_size, first_batch=first)


def _source(loader, trainer):
    """What this rank's feeder iterates: the loader itself on one GPU, its sharded view under a launcher (one view per
    loader, kept with the trainer so that the feeder cache — keyed by the object it iterates — finds it again)."""
    import weakref
    rank, world = world_info()
    if world == 1 or getattr(loader, "plb_per_rank", False):   # (build_dataloader(shard=...): already this rank's samples)
        return loader
    views = trainer.__dict__.setdefault("_sharded_views", weakref.WeakKeyDictionary())
    v = views.get(loader)
    if v is None:
        v = views[loader] = _ShardedLoader(loader, rank, world)
    return v


def _batches(loader, trainer, device_masking, word_separator, budget=None):
    """Collated batches of the loader as device-resident StagedBatch objects: this rank's contiguous slice of every
    global batch (accelerate's split_batches=True), copies overlapped with compute (DeviceFeeder) — on one GPU and, since
    round 5, on every rank of a data-parallel run (it staged each batch with blocking copies before).
    ``budget``: see DeviceFeeder.draw_budget (the training loader: draws are granted up to the next validation)."""
    yield from _feeder(_source(loader, trainer), trainer, word_separator, budget)


MAX_HANDOFF_RETRIES = 3


def _checked(trainer, call, note=None):
    """``float(call().item())`` with the engine's hand-off status checked right behind the read-back, where it is exact. A
    step the engine declared invalid (HandoffTimeout: NaN loss, update left out on every rank, engine reset — never
    observed) is run again, up to MAX_HANDOFF_RETRIES times; in a data-parallel run every rank sees the failure at
    the same step (the health word is agreed inside the step), so all ranks retry the same batch together."""
    from .engine import HandoffTimeout
    for attempt in range(MAX_HANDOFF_RETRIES + 1):
        try:
            loss = float(call().item())
            trainer.engine.raise_if_failed()
            return loss
        except HandoffTimeout as ex:
            if attempt == MAX_HANDOFF_RETRIES:
                raise
            if note is not None:
                note(handoff_timeout=str(ex), retry=attempt + 1)


def validate(trainer, val_loader, device_masking=False, word_separator=None):
    """Mean of the per-batch losses, forward only (train.py:288-304; masks are re-drawn each pass, as there)."""
    total, n = 0.0, 0
    for b in _batches(val_loader, trainer, device_masking, word_separator):
        total += _checked(trainer, lambda: trainer.engine.loss_fwd(b.masked, b.labels, b.lengths, b.offsets, b.flat, b.n_masked))
        n += 1
    return total / max(n, 1)


class _LossReader:
    """Read a step's loss back WITHOUT stalling the step behind it: ``post`` copies the engine's one-element loss buffer
    (overwritten by the next step) into one of two pinned host words on the step's stream and records an event;
    ``read`` waits for that event only — by then the next step is already enqueued, so the GPU never waits for the host
    to come back from a read-back (measured with the real pipeline, 4 workers, profiles/r05_train_loop_loss_readback_*.jsonl:
    9.57 -> 9.24 ms per step in bf16, 8.04 -> 7.77 in fp8 — the fed step without any read-back: 9.25 / 7.79)."""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device
        self.host = [torch.zeros(1).pin_memory() for _ in range(2)]
        self.events = [torch.cuda.Event() for _ in range(2)]
        self.n = 0

    def post(self, loss_dev):
        k = self.n & 1
        self.n += 1
        self.host[k].copy_(loss_dev, non_blocking=True)
        self.events[k].record(self.torch.cuda.current_stream(self.device))
        return k

    def read(self, k):
        self.events[k].synchronize()
        return float(self.host[k][0])


def train_loop(trainer, train_loader, val_loader, current_step, num_steps, save_interval, log_interval, log, log_dir,
               device_masking=False, word_separator=None, max_epochs=MAX_EPOCHS, deferred_readback=True, reader=None):
    """train.py:338-379: validation first, then epochs until ``num_steps``; checkpoint + validation every
    ``save_interval`` steps; every rank logs its LOCAL loss (the reference's rank 0 does).

    ``deferred_readback`` (default): the loss of step i is read back after step i + 1 has been enqueued (``_LossReader``) —
    the same records in the same order as the reference's ``loss.item()`` per step (train.py:395), without its stall. Steps
    that end an interval (checkpoint + validation next) or the run are read back at once, so those happen on exactly the
    weights the reference has there. A step the engine declared invalid (HandoffTimeout; never observed) is found when its
    loss is read — exactly, and at the same step on every rank — and it and the step enqueued behind it (which the device
    left out too: the word is sticky) are run again in order."""
    from .engine import HandoffTimeout
    main = world_info()[0] == 0
    window = deque(maxlen=log_interval)
    epoch = 0
    log(val_phoneme_loss=validate(trainer, val_loader, device_masking, word_separator), step=current_step, epoch=epoch)
    train_feeder = _feeder(_source(train_loader, trainer), trainer, word_separator, 0)

    def grant():
        # the training producer may draw up to the next validation point and no further: the validation pass then sees the
        # global masking streams exactly where the reference's single-threaded loop leaves them (train.py:369-373) — on
        # every rank of a data-parallel run alike (each rank draws the global batch's masks, as the reference's ranks do)
        train_feeder.grant(save_interval - current_step % save_interval)

    train_feeder.reset_budget(0)   # permits an earlier loop with this trainer and loader left unused (it stopped mid-interval)
    grant()
    if reader is None:
        reader = _LossReader(trainer.engine.device)
    note = lambda **kw: log(step=current_step, epoch=epoch, **kw)

    def record(loss):
        nonlocal current_step
        current_step += 1
        window.append(loss)
        rec = {"phoneme_loss": loss, "epoch": epoch, "step": current_step}
        if len(window) == log_interval:
            rec["phoneme_loss_avg"] = float(np.mean(window))
        log(**rec)

    def finish(p):
        """Read back and log the pending step p = (handle, batch). True: the engine has declared a step invalid — this one
        (its loss is NaN: it is run again here, checked) or one enqueued behind it that has completed already (this one's
        loss is a number) — and whatever was enqueued behind this step was left out by the device: enqueue it again."""
        import math
        loss = reader.read(p[0])
        redo = False
        try:
            trainer.engine.raise_if_failed()     # this step has completed: the word covers it (and may cover a later one)
        except HandoffTimeout as ex:
            note(handoff_timeout=str(ex), retry=1)
            if math.isnan(loss):
                loss = _checked(trainer, lambda: trainer.step(p[1]), note)
            redo = True
        record(loss)
        return redo

    pending = None                               # a step whose loss has not been read back yet
    while epoch < max_epochs:
        epoch += 1
        for batch in _batches(train_loader, trainer, device_masking, word_separator, 0):
            ahead = current_step + (pending is not None)        # steps enqueued before this one
            try:
                this = (reader.post(trainer.step(batch)), batch)
            except HandoffTimeout as ex:         # the pending step completed, invalid, before this one was enqueued
                note(handoff_timeout=str(ex), retry=1)
                record(_checked(trainer, lambda: trainer.step(pending[1]), note))
                pending = None
                this = (reader.post(trainer.step(batch)), batch)
            if pending is not None and finish(pending):
                this = (reader.post(trainer.step(batch)), batch)
            pending = this
            if not deferred_readback or (ahead + 1) % save_interval == 0 or ahead + 1 >= num_steps:
                finish(pending)                  # (nothing is enqueued behind it: an invalid step is simply run again)
                pending = None
                if current_step % save_interval == 0:
                    save_checkpoint(trainer, current_step, log_dir, epoch, main)
                    log(val_phoneme_loss=validate(trainer, val_loader, device_masking, word_separator), step=current_step,
                        epoch=epoch)
                    grant()
                if current_step >= num_steps:
                    return current_step, epoch
        if pending is not None:                  # the epoch's last step
            finish(pending)
            pending = None
    return current_step, epoch


def train(args=None, dataset=None, device=None):
    """Drop-in for ``train.train(args)`` (train.py:133-172): ``args = {'config_path', 'run_name'}``. ``dataset``: rows of
    ``{'phonemes': [...]}``; None loads ``training_params.training_dataset`` with HF ``datasets`` as the reference does."""
    args = args or parse_args()
    # under torchrun / accelerate launch: join the group and take this rank's GPU BEFORE anything touches a device or the
    # run directory (the reference gets both from Accelerator(), train.py:218-221)
    _, _, env_device = init_from_env()
    device = device or env_device
    config, log_dir, resuming = setup_config_and_directories(args, args["config_path"])
    tp, dp = config["training_params"], dict(config["dataset_params"])
    main = world_info()[0] == 0
    print(f"Resuming training from '{log_dir}' with existing config." if resuming else f"Starting new training run in '{log_dir}'.")
    if dataset is None:
        from datasets import load_dataset
        dataset = load_dataset(tp["training_dataset"], split=tp["split"])
    device_masking = bool(tp.get("device_masking", False))
    _, world = world_info()
    # training_params.shard_samples (not in the reference's config.yml): each rank loads and masks only its own samples
    # instead of the whole global batch (build_dataloader(shard=...)); the device-side masking then works under a launcher too
    shard = (world_info()[0], world) if world > 1 and bool(tp.get("shard_samples", False)) else None
    decisions = device_masking and (world == 1 or shard is not None)
    train_loader, val_loader = build_dataloader(dataset, batch_size=int(tp["batch_size"]), device="cuda", dataset_config=dp,
                                                use_token_ids=False, num_workers=int(tp.get("num_workers", 0)),
                                                decisions=decisions, shard=shard)
    trainer, current_step = initialize_model(config, log_dir, resuming, device=device)
    log = _Log(log_dir, main)
    print("Start training...")
    current_step, epoch = train_loop(trainer, train_loader, val_loader, current_step, int(tp["num_steps"]),
                                     int(tp["save_interval"]), int(tp["log_interval"]), log, log_dir,
                                     decisions, dp.get("word_separator"))
    print(f"Training completed at step {current_step}, epoch {epoch}")
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    return trainer, current_step, epoch


if __name__ == "__main__":
    train()
