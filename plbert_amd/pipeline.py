"""Host -> device input pipeline of the hot path (second half).

The reference feeds its step from ``DataLoader(num_workers=0)`` (train.py:253): Python string masking on the training
process's own thread, then synchronous ``.to(device)`` copies. ``DeviceFeeder`` keeps the GPU fed instead:

  worker processes (``build_dataloader(num_workers=N)``: independent masking streams, data.seed_worker)
    -> a producer thread packs every collated batch into ONE pinned host buffer
    -> ONE asynchronous copy on a copy stream into one of ``depth`` device slots (the copy of batch i+1 runs beside
       the step of batch i)
    -> the step's stream waits for the slot's event only; a slot is reused once the step that read it has been enqueued
       and its completion event recorded.

Two batch forms: collated tensors of PhonemeOnlyCollater / Collater (``labels, masked, lengths, indices``), or
``collate_decisions`` dicts, for which the cropping / substitution / index lists are produced on the GPU by
plb_apply_mask (bit-exact with the host path, tests/test_gpu_apply_mask.py)."""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

from .data import masked_indices_to_csr
from .train import StagedBatch, validate_batch


def _pack(arrays):
    """[(name, ndarray)] -> (uint8 array, {name: (offset, dtype, shape)}) with 16-byte aligned segments."""
    meta, off = {}, 0
    for name, a in arrays:
        a = np.ascontiguousarray(a)
        meta[name] = (off, a.dtype, a.shape)
        off += (a.nbytes + 15) // 16 * 16
    buf = np.empty(max(off, 16), dtype=np.uint8)
    for name, a in arrays:
        o = meta[name][0]
        a = np.ascontiguousarray(a)
        buf[o:o + a.nbytes] = a.view(np.uint8).ravel()
    return buf, meta


_TORCH_DT = {np.dtype("int64"): torch.int64, np.dtype("int32"): torch.int32, np.dtype("int8"): torch.int8}


class DeviceFeeder:
    """Iterate a DataLoader as device-resident ``StagedBatch`` objects, copies overlapped with compute.

    ``for batch in DeviceFeeder(loader, device): trainer.step(batch)``. Call ``release(batch)`` after enqueuing the
    work that reads a batch if you hold more than one batch at a time; plain iteration releases the previous batch
    when the next one is requested.

    ``draw_budget`` (None = unlimited): how many batches the producer thread may draw from the loader; ``grant(n)`` adds
    n. The reference's masking draws the process-global ``random`` / ``numpy.random`` streams inside ``__getitem__`` and
    runs validation BETWEEN two training batches (train.py:369-373), so a producer that prefetches past a validation
    point — or runs beside the validation loader's producer — changes which random numbers each batch sees, run to run.
    A caller that validates mid-epoch grants exactly the batches up to the next validation (``run.train_loop``), which
    reproduces the reference's draw order. One feeder object may be iterated many times (epochs, repeated validation): its
    pinned buffers, device slots and copy stream persist."""

    def __init__(self, loader, device=None, depth=2, vocab_size=None, validate=True, word_separator=None, prefetch=4,
                 draw_budget=None, mask_on_copy_stream=True):
        self.loader = loader
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.depth = max(2, int(depth))
        self.vocab_size, self.validate, self.word_separator = vocab_size, validate, word_separator
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.prefetch = prefetch
        self._slots = [None] * self.depth          # device uint8 buffers
        self._pinned = [None] * self.depth
        self._ready = [torch.cuda.Event() for _ in range(self.depth)]
        self._free = [torch.cuda.Event() for _ in range(self.depth)]
        self._prev = None
        # decision records: plb_apply_mask runs on the COPY stream right behind the upload (one batch ahead of the step that
        # reads it), and the masked count comes back through a pinned word on that stream — the step's stream never waits
        # for the host and the host never waits for the step (on the compute stream the count's read-back is queued behind
        # the previous step: the host then cannot enqueue a step before the one before it has finished)
        self.mask_on_copy_stream = bool(mask_on_copy_stream)
        self._outs = [None] * self.depth            # per slot: device outputs of plb_apply_mask + pinned count
        self._limited = draw_budget is not None
        self._budget = threading.Semaphore(int(draw_budget or 0))

    def grant(self, n):
        """Allow the producer to draw ``n`` more batches (only meaningful with ``draw_budget``)."""
        for _ in range(int(n)):
            self._budget.release()

    def reset_budget(self, n=0):
        """Forget permits a previous loop left unused (it returned in the middle of an interval) and start from ``n``:
        the producer must never draw past the NEXT validation point of the loop that is starting now."""
        while self._budget.acquire(blocking=False):
            pass
        self.grant(n)

    # ---- producer thread: DataLoader -> packed host arrays -------------------------------------------------
    def _put(self, q, item, stop):
        while not stop.is_set():  # a consumer that left early (break, exception) must not strand this thread
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    def _produce(self, q, stop):
        try:
            it = iter(self.loader)
            while True:
                if self._limited:
                    while not self._budget.acquire(timeout=0.2):
                        if stop.is_set():
                            return
                try:
                    item = next(it)
                except StopIteration:
                    if self._limited:
                        self._budget.release()      # nothing was drawn: the unit belongs to the next epoch's first batch
                    break
                if not self._put(q, self._host_pack(item), stop):
                    return
            self._put(q, None, stop)
        except BaseException as ex:  # surfaced in the consumer
            self._put(q, ex, stop)

    def _host_pack(self, item):
        if isinstance(item, dict):  # collate_decisions
            arrs = [(k, item[k]) for k in ("ids", "repl", "sample_off", "word_off", "word_begin", "word_len", "action",
                                           "crop_start")]
            if item["word_token"] is not None:
                arrs.append(("word_token", item["word_token"]))
            buf, meta = _pack(arrs)
            return ("decisions", buf, meta, dict(B=item["B"], S=item["S"], lengths=item["lengths"]))
        token_ids = None
        if len(item) == 5:
            token_ids, *item = item
        labels, masked, lengths, idx = item
        labels, masked = np.asarray(labels), np.asarray(masked)
        if self.validate and self.vocab_size:
            validate_batch(labels, masked, lengths, idx, self.vocab_size)
        off, flat = masked_indices_to_csr(idx)
        arrs = [("labels", labels.astype(np.int64)), ("masked", masked.astype(np.int64)), ("offsets", off), ("flat", flat),
                ("lengths", np.asarray(lengths, dtype=np.int32))]
        if token_ids is not None:
            arrs.append(("token_ids", np.asarray(token_ids).astype(np.int64)))
        buf, meta = _pack(arrs)
        return ("collated", buf, meta, dict(B=labels.shape[0], S=labels.shape[1], lengths=[int(x) for x in lengths],
                                            n_masked=int(off[-1])))

    # ---- consumer side --------------------------------------------------------------------------------------
    def _view(self, dev, meta, name):
        o, dt, shape = meta[name]
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        return dev[o:o + n].view(_TORCH_DT[np.dtype(dt)]).view(*shape) if n else torch.empty(shape, dtype=_TORCH_DT[np.dtype(dt)], device=self.device)

    def _upload(self, k, buf):
        n = buf.nbytes
        if self._pinned[k] is None or self._pinned[k].numel() < n:
            cap = max(n * 2, 1 << 16)
            self._pinned[k] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            cur = torch.cuda.current_stream(self.device)
            if self._slots[k] is not None:
                self._slots[k].record_stream(cur)   # the step still reading the old slot keeps its memory until it is done
            # the slot belongs to the COPY stream's allocator pool: allocated under the compute stream it could alias a
            # block that a step still in flight has just freed, and the copy below does not wait for that step
            with torch.cuda.stream(self.copy_stream):
                self._slots[k] = torch.empty(cap, dtype=torch.uint8, device=self.device)
            self._slots[k].record_stream(cur)       # read by the compute stream from now on
        self._free[k].synchronize()                 # the step that last read this slot (and its pinned twin) is done
        self._pinned[k][:n].numpy()[:] = buf
        with torch.cuda.stream(self.copy_stream):
            self._slots[k][:n].copy_(self._pinned[k][:n], non_blocking=True)
            self._ready[k].record(self.copy_stream)
        return self._slots[k]

    def release(self, batch):
        k = getattr(batch, "_slot", None)
        if k is not None:
            self._free[k].record(torch.cuda.current_stream(self.device))
            batch._slot = None

    def __iter__(self):
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        th = threading.Thread(target=self._produce, args=(q, stop), daemon=True)
        th.start()
        try:
            yield from self._consume(q)
        finally:
            stop.set()
            if self._prev is not None:
                self.release(self._prev)
                self._prev = None
            th.join(timeout=5)

    def _consume(self, q):
        for k in range(self.depth):
            self._free[k].record(torch.cuda.current_stream(self.device))
        pending = []                                # uploaded, not yet yielded: (slot, kind, meta, info)
        i, done = 0, False
        with torch.cuda.device(self.device):
            while True:
                while not done and len(pending) < self.depth - 1:
                    # With a draw budget the producer may be parked on it (grant() only comes after this consumer has
                    # yielded): block for the batch that is needed NOW, take further ones only if they are already there
                    # (depth >= 3 would otherwise wait for a batch beyond the budget: a deadlock at every validation point)
                    try:
                        item = q.get(block=not (self._limited and pending))
                    except queue.Empty:
                        break
                    if item is None:
                        done = True
                        break
                    if isinstance(item, BaseException):
                        raise item
                    kind, buf, meta, info = item
                    k = i % self.depth
                    i += 1
                    dev = self._upload(k, buf)
                    if kind == "decisions" and self.mask_on_copy_stream:
                        self._apply_mask(k, meta, info, dev, self.copy_stream)
                    pending.append((k, kind, meta, info, dev))
                if not pending:
                    break
                if self._prev is not None:
                    self.release(self._prev)
                k, kind, meta, info, dev = pending.pop(0)
                torch.cuda.current_stream(self.device).wait_event(self._ready[k])
                info["_slot"] = k
                batch = self._stage(kind, meta, info, dev)
                batch._slot = k
                self._prev = batch
                yield batch

    def _stage(self, kind, meta, info, dev):
        B, S, lengths = info["B"], info["S"], info["lengths"]
        if kind == "collated":
            lens_t = None if all(x == S for x in lengths) else self._view(dev, meta, "lengths")
            tok = self._view(dev, meta, "token_ids") if "token_ids" in meta else None
            return StagedBatch(self._view(dev, meta, "masked"), self._view(dev, meta, "labels"), lens_t,
                               self._view(dev, meta, "offsets"), self._view(dev, meta, "flat"), info["n_masked"],
                               int(sum(lengths)), tok)
        # decisions: the masking is applied on the device (plb_apply_mask)
        k = info["_slot"]
        if not self.mask_on_copy_stream:
            self._apply_mask(k, meta, info, dev, torch.cuda.current_stream(self.device))
        o = self._outs[k]
        o["done"].synchronize()                     # the masked count has landed in the pinned word (copy stream: long ago)
        n = int(o["n_host"][0])                     # the one host read of the path: the masked count sizes the loss GEMM
        v = lambda name: o[name][: B * S].view(B, S)
        lens_t = None if all(x == S for x in lengths) else o["lens"][:B]
        return StagedBatch(v("masked"), v("labels"), lens_t, o["offsets"][: B + 1], o["flat"][:n], n, int(sum(lengths)),
                           v("tokens") if "word_token" in meta else None)

    def _apply_mask(self, k, meta, info, dev, stream):
        """plb_apply_mask of slot k's decision records on ``stream`` (behind the upload), outputs into the slot's own
        device buffers (kept for the feeder's lifetime; grown under the copy stream's allocator pool), the masked count
        copied into a pinned word; ``_ready[k]`` / ``done`` re-recorded behind it."""
        import ctypes as C
        from . import _lib
        from .symbols import MASK_ID
        L = _lib.lib()
        B, S = info["B"], info["S"]
        o = self._outs[k]
        if o is None or o["cap"] < B * S or o["bcap"] < B:
            cap, bcap = max(B * S, o["cap"] if o else 0), max(B, o["bcap"] if o else 0)
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                i64 = lambda n: torch.empty(n, dtype=torch.int64, device=self.device)
                i32 = lambda n: torch.empty(n, dtype=torch.int32, device=self.device)
                new = dict(cap=cap, bcap=bcap, labels=i64(cap), masked=i64(cap), tokens=i64(cap), lens=i32(bcap),
                           offsets=i32(bcap + 1), flat=i32(cap), scratch=i32(bcap + cap),
                           n_host=torch.zeros(1, dtype=torch.int32).pin_memory(), done=torch.cuda.Event())
            for t in new.values():
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
            if o is not None:
                for t in o.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)        # a step still reading the old buffers keeps them until it is done
            o = self._outs[k] = new
        v = lambda n: self._view(dev, meta, n)
        wtok = v("word_token") if "word_token" in meta else None
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.stream(stream):
            if stream is not self.copy_stream:
                stream.wait_event(self._ready[k])
            _lib.check(L.plb_apply_mask(v("ids").data_ptr(), v("sample_off").data_ptr(), v("word_off").data_ptr(),
                                        v("word_begin").data_ptr(), v("word_len").data_ptr(), v("action").data_ptr(),
                                        v("repl").data_ptr(), p(wtok), int(self.word_separator or 0), v("crop_start").data_ptr(),
                                        B, S, MASK_ID, o["labels"].data_ptr(), o["masked"].data_ptr(),
                                        o["tokens"].data_ptr() if wtok is not None else None, o["lens"].data_ptr(),
                                        o["offsets"].data_ptr(), o["flat"].data_ptr(), o["scratch"].data_ptr(),
                                        C.c_void_p(stream.cuda_stream)), "plb_apply_mask")
            o["n_host"].copy_(o["offsets"][B:B + 1], non_blocking=True)
            o["done"].record(stream)
            if stream is self.copy_stream:
                self._ready[k].record(stream)
