"""Host side of the HIP engine: device buffers (torch tensors as containers) + the C-ABI calls.

One ``HipEngine`` = one GPU = one set of flat fp32 parameter / gradient / AdamW-moment buffers laid
out as ``plb_param_layout`` says, plus the zero-filled workspace the native engine carves up.
PyTorch supplies memory, streams and (in ``train.py``) ``torch.distributed``; all arithmetic of the
hot path runs in libplbert_hip.so.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .init import param_shapes


class HandoffTimeout(RuntimeError):
    """A fused LayerNorm launch's in-launch hand-off timed out (include/plbert.h: plb_status): the step it belongs to is
    invalid. By the time this is raised the engine has skipped that step's optimizer update and has been reset
    (exchange buffer and error word zeroed): the caller may simply run the step again."""


class HipEngine:
    def __init__(self, cfg, num_phonemes, num_tokens=0, max_batch=32, max_seq=512, device=None, train=True):
        """``train=False``: inference / validation engine (README.md:91, train.py:288-304) — no gradient, moment or
        per-layer activation buffers (forward and loss-only calls work, backward and AdamW raise)."""
        cfg.check_supported()
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "the PL-BERT hot path has no CPU fallback")
        self.cfg = cfg
        self.num_phonemes, self.num_tokens = int(num_phonemes), int(num_tokens)
        self.max_batch, self.max_seq = int(max_batch), int(max_seq)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.train_mode = bool(train)
        self.L = _lib.lib()
        c = _lib.PlbConfig(cfg.vocab_size, cfg.embedding_size, cfg.hidden_size, cfg.num_attention_heads,
                           cfg.intermediate_size, cfg.num_hidden_layers, cfg.max_position_embeddings,
                           cfg.type_vocab_size, cfg.layer_norm_eps, self.num_phonemes, self.num_tokens,
                           self.max_batch, self.max_seq, 0 if train else 1)
        h = C.c_void_p()
        _lib.check(self.L.plb_create(C.byref(c), C.byref(h)), "plb_create")
        self.handle = h
        offs = (C.c_int64 * _lib.PLB_NPARAM)()
        sizes = (C.c_int64 * _lib.PLB_NPARAM)()
        total, trainable = C.c_int64(), C.c_int64()
        _lib.check(self.L.plb_param_layout(h, offs, sizes, C.byref(total), C.byref(trainable)), "plb_param_layout")
        self.total, self.trainable = int(total.value), int(trainable.value)
        shapes = param_shapes(cfg, self.num_phonemes, self.num_tokens)
        self.layout = OrderedDict()
        for i, name in enumerate(_lib.PLB_PARAM_NAMES):
            if sizes[i] == 0:
                continue
            shp = shapes[name]
            assert int(np.prod(shp)) == sizes[i], (name, shp, sizes[i])
            self.layout[name] = (int(offs[i]), int(sizes[i]), tuple(shp))
        # every C-ABI call below runs with this engine's device current (plb_bind creates its side stream and events
        # there; launches go to that device's streams): HipEngine(device="cuda:1") works while cuda:0 is current
        with torch.cuda.device(self.device):
            z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
            self.params = z(self.total)
            self._loss = z(1)
            self._loss_parts = z(2)
        # gradients, AdamW moments and the workspace are allocated by the first call that computes (_bind): a model
        # that is only constructed, or whose parameters move into another engine (model.py: _adopt), holds its
        # parameters and nothing else
        self._grads = self._exp_avg = self._exp_avg_sq = self.workspace = None
        self.ws_bytes = int(self.L.plb_workspace_bytes(h))
        self._bound = False
        self._synced_version = -1
        self.comm_world = 1
        self._on_handoff_timeout = []   # callbacks(err): owners of an optimizer step count rewind it by err.skipped_updates

    def _bind(self):
        if self._bound:
            return
        with torch.cuda.device(self.device):
            z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
            if self.train_mode:
                self._grads, self._exp_avg, self._exp_avg_sq = z(self.total), z(self.total), z(self.total)
            self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)
            p = lambda t: None if t is None else t.data_ptr()
            # plb_bind creates the engine's side stream and events on the current device
            _lib.check(self.L.plb_bind(self.handle, self.params.data_ptr(), p(self._grads), p(self._exp_avg),
                                       p(self._exp_avg_sq), self.workspace.data_ptr(), self.ws_bytes), "plb_bind")
        self._bound = True

    # flat fp32 gradient / AdamW-moment buffers (None on an inference engine); first access allocates
    @property
    def grads(self):
        if self.train_mode:
            self._bind()
        return self._grads

    @property
    def exp_avg(self):
        if self.train_mode:
            self._bind()
        return self._exp_avg

    @property
    def exp_avg_sq(self):
        if self.train_mode:
            self._bind()
        return self._exp_avg_sq

    def device_bytes(self):
        """Bytes of device memory this engine holds."""
        ts = [self.params, self._grads, self._exp_avg, self._exp_avg_sq, self.workspace]
        return sum(t.numel() * t.element_size() for t in ts if t is not None)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.L.plb_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------------------
    def view(self, name, of=None):
        off, n, shp = self.layout[name]
        return (self.params if of is None else of)[off:off + n].view(shp)

    def load_state_dict(self, sd, strict=True):
        """Copy tensors / arrays keyed by reference state-dict names into the flat buffer.
        ``module.`` prefixes are stripped as train.py:98 does."""
        seen = set()
        for k, v in sd.items():
            k = k.replace("module.", "")
            if k not in self.layout:
                if strict and not (k.endswith("position_ids") or k.endswith("token_type_ids")):
                    raise KeyError(f"unexpected key {k}")
                continue
            t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
            self.view(k).copy_(t.to(self.device, torch.float32))
            seen.add(k)
        if strict:
            missing = [k for k in self.layout if k not in seen]
            if missing:
                raise KeyError(f"missing keys {missing}")
        self._synced_version = -1  # the compute copies are refreshed by the next call that computes

    def state_dict(self):
        return OrderedDict((k, self.view(k).detach().clone()) for k in self.layout)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def sync_weights(self):
        """Refresh the bf16 / transposed compute copies from the fp32 parameters. Automatic after load_state_dict,
        AdamW and any in-place op torch tracks on the parameters; call it yourself after writing through ``.data``
        (which bypasses torch's version counter)."""
        self._bind()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_sync_weights(self.handle, self._stream()), "plb_sync_weights")
        self._synced_version = self.params._version

    def _ensure_synced(self):
        self._bind()
        if self.params._version != self._synced_version:
            self.sync_weights()

    # ---- data-parallel exchange (plb_comm_*) ------------------------------------------------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """Attach an RCCL communicator (collective: every rank calls it with rank 0's ``comm_unique_id()``)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._bind()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_comm_init(self.handle, buf, int(rank), int(world)), "plb_comm_init")
        self.comm_world = int(world)

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().plb_comm_unique_id(buf), "plb_comm_unique_id")
        return bytes(buf)

    def comm_info(self):
        r, w, v = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.L.plb_comm_info(self.handle, C.byref(r), C.byref(w), C.byref(v)), "plb_comm_info")
        return int(r.value), int(w.value), int(v.value)

    def status(self):
        """{'ln_exchange_timeouts': n} — synchronises the device (plb_status); a non-zero report resets the exchange
        state, so it is reported once."""
        n, k = C.c_int32(), C.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_status_ex(self.handle, C.byref(n), C.byref(k)), "plb_status_ex")
        return {"ln_exchange_timeouts": int(n.value), "skipped_updates": int(k.value)}

    def poll_status(self):
        """The same count WITHOUT synchronising: as of the last loss call that has completed on the device."""
        if not self._bound:
            return {"ln_exchange_timeouts": 0}
        n = C.c_int32()
        _lib.check(self.L.plb_poll_status(self.handle, C.byref(n)), "plb_poll_status")
        return {"ln_exchange_timeouts": int(n.value)}

    def raise_if_failed(self):
        """Raise HandoffTimeout if a completed step reported a timed-out hand-off. Costs one read of a pinned host
        word: called at the top of every training step (one step late at worst — the device has already left the
        failed step's update out) and wherever a loss has just been read back (exact)."""
        if self.poll_status()["ln_exchange_timeouts"]:
            st = self.status()   # synchronises, reports once and resets the exchange state
            err = HandoffTimeout(f"{st['ln_exchange_timeouts']} in-launch LayerNorm hand-off(s) timed out: the loss of "
                                 f"that step is NaN and {st['skipped_updates']} optimizer update(s) were skipped on the "
                                 "device; the engine has been reset, re-run the step")
            err.skipped_updates = st["skipped_updates"]
            for fn in self._on_handoff_timeout:
                fn(err)
            raise err

    def status_exchange(self, all_reduce):
        """For a host-side gradient exchange (torch.distributed fallback): let the step's health word travel with the
        gradients — export this rank's count, ``all_reduce(tensor)`` sums it over the ranks in place, import merges the
        sum (every rank then skips the update, returns a NaN loss and raises HandoffTimeout, or none does). With the
        engine's own communicator the same happens inside plb_loss_fwd_bwd."""
        if not hasattr(self, "_status_f"):
            self._status_f = torch.zeros(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_status_export(self.handle, self._status_f.data_ptr(), self._stream()), "plb_status_export")
            all_reduce(self._status_f)
            _lib.check(self.L.plb_status_import(self.handle, self._status_f.data_ptr(), self._stream()), "plb_status_import")

    def hb_audit(self, on=True, break_wait=-1):
        """Debug: happens-before audit of the backward's streams (include/plbert.h: plb_debug_hb_audit)."""
        _lib.check(self.L.plb_debug_hb_audit(self.handle, int(bool(on)), int(break_wait)), "plb_debug_hb_audit")

    def hb_report(self):
        n, v = C.c_int64(), C.c_int32()
        buf = C.create_string_buffer(512)
        _lib.check(self.L.plb_debug_hb_report(self.handle, C.byref(n), C.byref(v), buf, 512), "plb_debug_hb_report")
        return {"checks": int(n.value), "violations": int(v.value), "first": buf.value.decode()}

    def comm_trace(self, on=True):
        _lib.check(self.L.plb_comm_trace(self.handle, int(bool(on))), "plb_comm_trace")

    def comm_trace_read(self, max_pieces=16):
        """Per piece of the last loss call: (begin, end) float range, release and completion time in ms since the call's
        first launch; plus (begin, end) of the weight-gradient tail. Synchronises on the trace events."""
        n = C.c_int32()
        a, b = (C.c_int64 * max_pieces)(), (C.c_int64 * max_pieces)()
        r, d, t = (C.c_float * max_pieces)(), (C.c_float * max_pieces)(), (C.c_float * 2)()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_comm_trace_read(self.handle, max_pieces, C.byref(n), a, b, r, d, t), "plb_comm_trace_read")
        return {"tail_ms": [round(t[0], 4), round(t[1], 4)],
                "pieces": [{"range": [int(a[i]), int(b[i])], "released_ms": round(r[i], 4), "done_ms": round(d[i], 4)}
                           for i in range(n.value)]}

    def last_application_rows(self):
        """(rows, of): token rows the last loss call ran the post-attention part of its last application on, of the
        call's padded token count (plb_last_application_rows)."""
        r, o = C.c_int64(), C.c_int64()
        _lib.check(self.L.plb_last_application_rows(self.handle, C.byref(r), C.byref(o)), "plb_last_application_rows")
        return int(r.value), int(o.value)

    def comm_pieces(self):
        """(collectives, floats) of the last step's gradient exchange."""
        n, f = C.c_int32(), C.c_int64()
        _lib.check(self.L.plb_comm_pieces(self.handle, C.byref(n), C.byref(f)), "plb_comm_pieces")
        return int(n.value), int(f.value)

    def comm_destroy(self):
        with torch.cuda.device(self.device):
            self.L.plb_comm_destroy(self.handle)
        self.comm_world = 1

    def set_grad_overlap(self, on):
        _lib.check(self.L.plb_set_grad_overlap(self.handle, int(bool(on))), "plb_set_grad_overlap")

    def broadcast_params(self, root=0):
        self._bind()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_broadcast_params(self.handle, int(root), self._stream()), "plb_broadcast_params")
        self._synced_version = self.params._version

    def allreduce_grads(self):
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_allreduce_grads(self.handle, self._stream()), "plb_allreduce_grads")

    def set_fp8(self, on=True):
        """fp8 (e4m3 / e5m2) MFMA path for the QKV / FFN GEMMs (plb_set_fp8); the next call calibrates in bf16."""
        self._bind()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_set_fp8(self.handle, int(bool(on)), self._stream()), "plb_set_fp8")

    def fp8_state(self):
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self.L.plb_fp8_state(self.handle, C.byref(a), C.byref(b)), "plb_fp8_state")
        return bool(a.value), bool(b.value)

    FP8_SITES = ("x", "a", "gelu_u", "context", "dpre2", "dU", "dpre1", "dQKV")

    def fp8_stats(self):
        """{site: (calls in which values were clamped, worst overshoot)} since set_fp8 (plb_fp8_stats; synchronises)."""
        c, w = (C.c_float * 8)(), (C.c_float * 8)()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_fp8_stats(self.handle, c, w, self._stream()), "plb_fp8_stats")
        return {n: (int(c[i]), round(float(w[i]), 4)) for i, n in enumerate(self.FP8_SITES)}

    @property
    def token_head_steps(self):
        return int(self.L.plb_token_head_steps(self.handle))

    @token_head_steps.setter
    def token_head_steps(self, n):
        _lib.check(self.L.plb_set_token_head_steps(self.handle, int(n)), "plb_set_token_head_steps")

    # ---- calls --------------------------------------------------------------------------------------------
    def _dev_i64(self, x):
        t = torch.as_tensor(x)
        if t.dtype != torch.int64 or t.device != self.device or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.int64, non_blocking=True).contiguous()
        return t

    def _dev_i32(self, x):
        if x is None:
            return None
        t = torch.as_tensor(x)
        if t.dtype != torch.int32 or t.device != self.device or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
        return t

    def forward(self, ids, lengths=None, want_hidden=False, want_phoneme=True, want_token=False):
        """ids int64 [B,S]; lengths per-sample valid-token counts (None = no padding).
        Returns (hidden | None, phoneme_logits | None, token_logits | None), fp32 on the device."""
        self._ensure_synced()
        ids = self._dev_i64(ids)
        B, S = ids.shape
        lens = self._dev_i32(lengths)
        with torch.cuda.device(self.device):
            hid = torch.empty((B, S, self.cfg.hidden_size), dtype=torch.float32, device=self.device) if want_hidden else None
            ph = torch.empty((B, S, self.num_phonemes), dtype=torch.float32, device=self.device) if want_phoneme else None
            tk = torch.empty((B, S, self.num_tokens), dtype=torch.float32, device=self.device) if want_token else None
            p = lambda t: None if t is None else t.data_ptr()
            _lib.check(self.L.plb_forward(self.handle, ids.data_ptr(), p(lens), B, S, p(hid), p(ph), p(tk), self._stream()),
                       "plb_forward")
        return hid, ph, tk

    def pooler(self, hidden):
        """tanh(pooler(hidden[:, 0])) (modeling_albert.py:403), fp32 [B,H]; ``hidden`` fp32 [B,S,H] on the device."""
        self._ensure_synced()
        B, S, H = hidden.shape
        with torch.cuda.device(self.device):
            out = torch.empty((B, H), dtype=torch.float32, device=self.device)
            _lib.check(self.L.plb_pooler(self.handle, hidden.data_ptr(), B, S, out.data_ptr(), self._stream()), "plb_pooler")
        return out

    def loss_fwd(self, masked_ids, labels, lengths, idx_offsets, idx_flat, n_masked, token_ids=None):
        """Loss of one batch WITHOUT the backward (validate(), train.py:288-304; process_batch under no_grad):
        plb_loss_fwd — the gradient buffer is not touched. Returns the 1-element device loss tensor."""
        return self._loss_call(False, masked_ids, labels, lengths, idx_offsets, idx_flat, n_masked, token_ids)

    def loss_fwd_bwd(self, masked_ids, labels, lengths, idx_offsets, idx_flat, n_masked, token_ids=None):
        """Loss of one batch + gradients of every trainable parameter into ``self.grads``.
        Returns the 1-element device tensor holding the loss (no host sync).
        With ``token_ids`` (int64 [B,S], the 4-tuple Collater's first element) the step is dual-head:
        loss = phoneme loss + token loss, ``self.loss_parts`` holds the two terms and the token head's
        gradients are produced too."""
        return self._loss_call(True, masked_ids, labels, lengths, idx_offsets, idx_flat, n_masked, token_ids)

    def _loss_call(self, backward, masked_ids, labels, lengths, idx_offsets, idx_flat, n_masked, token_ids):
        if backward and not self.train_mode:
            raise RuntimeError("this HipEngine was built with train=False (inference / validation only)")
        self.raise_if_failed()   # a step that completed since the last call and timed out: never train on top of it
        self._ensure_synced()
        masked_ids = self._dev_i64(masked_ids)
        labels = self._dev_i64(labels)
        B, S = masked_ids.shape
        lens = self._dev_i32(lengths)
        offs = self._dev_i32(idx_offsets)
        flat = self._dev_i32(idx_flat)
        tok = None
        if token_ids is not None:
            if not self.num_tokens:
                raise ValueError("token_ids given but the engine was built without a token head (num_tokens = 0)")
            tok = self._dev_i64(token_ids)
            if tok.shape != masked_ids.shape:
                raise ValueError("token_ids must have the shape of the phoneme batch")
        p = lambda t: None if t is None else t.data_ptr()
        flat_p = flat.data_ptr() if n_masked else None
        with torch.cuda.device(self.device):
            if not backward:
                _lib.check(self.L.plb_loss_fwd(self.handle, masked_ids.data_ptr(), labels.data_ptr(), p(tok), p(lens),
                                               offs.data_ptr(), flat_p, int(n_masked), B, S, self._loss.data_ptr(),
                                               self._loss_parts.data_ptr() if tok is not None else None, self._stream()),
                           "plb_loss_fwd")
            elif tok is not None:
                _lib.check(self.L.plb_loss_fwd_bwd_dual(self.handle, masked_ids.data_ptr(), labels.data_ptr(), tok.data_ptr(),
                                                        p(lens), offs.data_ptr(), flat_p, int(n_masked), B, S,
                                                        self._loss.data_ptr(), self._loss_parts.data_ptr(), self._stream()),
                           "plb_loss_fwd_bwd_dual")
            else:
                _lib.check(self.L.plb_loss_fwd_bwd(self.handle, masked_ids.data_ptr(), labels.data_ptr(), p(lens),
                                                   offs.data_ptr(), flat_p, int(n_masked), B, S, self._loss.data_ptr(),
                                                   self._stream()), "plb_loss_fwd_bwd")
        return self._loss

    @property
    def loss_parts(self):
        """(phoneme loss, token loss) of the last dual-head call, on the device."""
        return self._loss_parts

    @property
    def token_range(self):
        """[start, end) of token_predictor.{weight,bias} in the flat buffers (empty without a token head)."""
        if not self.num_tokens:
            return (self.total, self.total)
        return (self.layout["token_predictor.weight"][0], self.total)

    def adamw_step(self, step, lr=7e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale=1.0):
        self._bind()
        with torch.cuda.device(self.device):
            _lib.check(self.L.plb_adamw_step(self.handle, lr, betas[0], betas[1], eps, weight_decay, int(step),
                                             grad_scale, self._stream()), "plb_adamw_step")
        self._synced_version = self.params._version
