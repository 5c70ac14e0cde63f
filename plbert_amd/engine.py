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


class HipEngine:
    def __init__(self, cfg, num_phonemes, num_tokens=0, max_batch=32, max_seq=512, device=None):
        cfg.check_supported()
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "the PL-BERT hot path has no CPU fallback")
        self.cfg = cfg
        self.num_phonemes, self.num_tokens = int(num_phonemes), int(num_tokens)
        self.max_batch, self.max_seq = int(max_batch), int(max_seq)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.L = _lib.lib()
        c = _lib.PlbConfig(cfg.vocab_size, cfg.embedding_size, cfg.hidden_size, cfg.num_attention_heads,
                           cfg.intermediate_size, cfg.num_hidden_layers, cfg.max_position_embeddings,
                           cfg.type_vocab_size, cfg.layer_norm_eps, self.num_phonemes, self.num_tokens,
                           self.max_batch, self.max_seq)
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
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.exp_avg = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.exp_avg_sq = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.ws_bytes = int(self.L.plb_workspace_bytes(h))
            self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)
            self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._loss_parts = torch.zeros(2, dtype=torch.float32, device=self.device)
        _lib.check(self.L.plb_bind(h, self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), self.workspace.data_ptr(), self.ws_bytes), "plb_bind")
        self._synced_version = -1

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
        self.sync_weights()

    def state_dict(self):
        return OrderedDict((k, self.view(k).detach().clone()) for k in self.layout)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def sync_weights(self):
        _lib.check(self.L.plb_sync_weights(self.handle, self._stream()), "plb_sync_weights")
        self._synced_version = self.params._version

    def _ensure_synced(self):
        if self.params._version != self._synced_version:
            self.sync_weights()

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

    def loss_fwd_bwd(self, masked_ids, labels, lengths, idx_offsets, idx_flat, n_masked, token_ids=None):
        """Loss of one batch + gradients of every trainable parameter into ``self.grads``.
        Returns the 1-element device tensor holding the loss (no host sync).
        With ``token_ids`` (int64 [B,S], the 4-tuple Collater's first element) the step is dual-head:
        loss = phoneme loss + token loss, ``self.loss_parts`` holds the two terms and the token head's
        gradients are produced too."""
        self._ensure_synced()
        masked_ids = self._dev_i64(masked_ids)
        labels = self._dev_i64(labels)
        B, S = masked_ids.shape
        lens = self._dev_i32(lengths)
        offs = self._dev_i32(idx_offsets)
        flat = self._dev_i32(idx_flat)
        if token_ids is not None:
            if not self.num_tokens:
                raise ValueError("token_ids given but the engine was built without a token head (num_tokens = 0)")
            tok = self._dev_i64(token_ids)
            if tok.shape != masked_ids.shape:
                raise ValueError("token_ids must have the shape of the phoneme batch")
            _lib.check(self.L.plb_loss_fwd_bwd_dual(self.handle, masked_ids.data_ptr(), labels.data_ptr(), tok.data_ptr(),
                                                    None if lens is None else lens.data_ptr(), offs.data_ptr(),
                                                    flat.data_ptr() if n_masked else None, int(n_masked), B, S,
                                                    self._loss.data_ptr(), self._loss_parts.data_ptr(), self._stream()),
                       "plb_loss_fwd_bwd_dual")
            return self._loss
        _lib.check(self.L.plb_loss_fwd_bwd(self.handle, masked_ids.data_ptr(), labels.data_ptr(),
                                           None if lens is None else lens.data_ptr(), offs.data_ptr(),
                                           flat.data_ptr() if n_masked else None, int(n_masked), B, S,
                                           self._loss.data_ptr(), self._stream()), "plb_loss_fwd_bwd")
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
        _lib.check(self.L.plb_adamw_step(self.handle, lr, betas[0], betas[1], eps, weight_decay, int(step),
                                         grad_scale, self._stream()), "plb_adamw_step")
        self._synced_version = self.params._version
