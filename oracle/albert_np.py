"""CPU oracle for the PL-BERT masked-phoneme pre-training step.  TEST INFRASTRUCTURE ONLY.

This file is the checker, not the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path (``plbert_amd/``) never
routes through it and fails loudly when the HIP library is missing.

It restates, in plain numpy, the arithmetic of the reference hot path:

* ``model.py:5-30``            MultiTaskModel / PhonemeOnlyModel heads
* ``train.py:34-44``           length_to_mask
* ``train.py:107-131``         calculate_phoneme_loss (mean of per-sample mean CE over non-empty samples)
* ``train.py:381-390``         process_batch
* ``train.py:272,355-357``     torch.optim.AdamW(lr) defaults, zero_grad/backward/step
* the third-party arithmetic the reference calls (HuggingFace ``transformers`` 5.15.0, not vendored
  in /root/reference, no version pinned there): ``models/albert/modeling_albert.py:49-106``
  (AlbertEmbeddings), ``:110-135`` (eager attention), ``:138-200`` (AlbertAttention), ``:203-238``
  (AlbertLayer), ``:258-284`` (AlbertTransformer, one shared layer applied L times), ``:372-408``
  (AlbertModel.forward), ``activations.py:59-66`` (gelu_new), config defaults
  ``configuration_albert.py:56-75``.

Pinning: the reference holds no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, captured in the build container by
``oracle/gen_golden.py`` and committed under ``tests/golden/`` (``tests/test_oracle_golden.py``).

Parameters are a dict keyed by the reference ``state_dict`` names (Linear weights ``[out, in]``).
All functions take a ``dtype`` (float32 by default; float64 to separate rounding from logic).
"""
from __future__ import annotations

import math

import numpy as np

ENC = "encoder."
LAYER = "encoder.encoder.albert_layer_groups.0.albert_layers.0."


def gelu_new(x):
    """HF NewGELUActivation (activations.py:59-66)."""
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + np.tanh(c * (x + 0.044715 * x * x * x)))


def gelu_new_grad(x):
    c = math.sqrt(2.0 / math.pi)
    t = np.tanh(c * (x + 0.044715 * x * x * x))
    return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * c * (1.0 + 3.0 * 0.044715 * x * x)


def layer_norm_fwd(x, g, b, eps):
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)  # biased, as torch.nn.LayerNorm
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * g + b, (xhat, rstd)


def layer_norm_bwd(dy, cache, g):
    xhat, rstd = cache
    red = tuple(range(dy.ndim - 1))
    dg = (dy * xhat).sum(red)
    db = dy.sum(red)
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx, dg, db


def length_to_mask(lengths):
    """train.py:34-44 — True on PAD positions: (pos + 1) > length."""
    lengths = np.asarray(lengths)
    max_len = int(lengths.max())
    pos = np.arange(max_len)[None, :]
    return (pos + 1) > lengths[:, None]


def attention_mask_from_lengths(lengths):
    """train.py:386 — (~text_mask).int(): 1 on valid tokens."""
    return (~length_to_mask(lengths)).astype(np.int32)


def _lin(x, w, b):
    return x @ w.T + b


class Config:
    """The subset of AlbertConfig the path uses (configuration_albert.py:56-75 defaults)."""

    def __init__(self, vocab_size=188, embedding_size=128, hidden_size=768, num_attention_heads=12,
                 intermediate_size=2048, num_hidden_layers=12, max_position_embeddings=512,
                 type_vocab_size=2, layer_norm_eps=1e-12, num_phonemes=188, num_tokens=0):
        self.vocab_size = vocab_size
        self.embedding_size = embedding_size
        self.hidden_size = hidden_size
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.layer_norm_eps = layer_norm_eps
        self.num_phonemes = num_phonemes
        self.num_tokens = num_tokens


def encoder_forward(cfg, P, ids, attention_mask=None, dtype=np.float32, keep=True):
    """AlbertModel.forward → last_hidden_state [B,S,H]  (modeling_albert.py:372-408).

    attention_mask: int [B,S], 1 = valid key (train.py:386), or None.  Only keys are masked; padded
    query rows still produce (finite) outputs, as in the reference.
    Returns (hidden, caches) — caches hold what ``encoder_backward`` needs.
    """
    ids = np.asarray(ids)
    B, S = ids.shape
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    eps = cfg.layer_norm_eps
    p = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
    caches = {"ids": ids, "layers": []}

    # AlbertEmbeddings (modeling_albert.py:67-106): word + token_type[0] + position[0..S)
    e_sum = (p[ENC + "embeddings.word_embeddings.weight"][ids]
             + p[ENC + "embeddings.token_type_embeddings.weight"][0][None, None, :]
             + p[ENC + "embeddings.position_embeddings.weight"][:S][None, :, :])
    e, c_eln = layer_norm_fwd(e_sum, p[ENC + "embeddings.LayerNorm.weight"],
                              p[ENC + "embeddings.LayerNorm.bias"], eps)
    caches["eln"] = c_eln
    caches["e"] = e
    # embedding_hidden_mapping_in (modeling_albert.py:272)
    x = _lin(e, p[ENC + "encoder.embedding_hidden_mapping_in.weight"],
             p[ENC + "encoder.embedding_hidden_mapping_in.bias"])

    if attention_mask is not None:
        am = np.asarray(attention_mask)
        bias = np.where(am[:, None, None, :] != 0, 0.0, np.finfo(dtype).min).astype(dtype)
        if (am != 0).all():
            bias = None  # create_bidirectional_mask returns None when nothing is padded
    else:
        bias = None

    Wq, bq = p[LAYER + "attention.query.weight"], p[LAYER + "attention.query.bias"]
    Wk, bk = p[LAYER + "attention.key.weight"], p[LAYER + "attention.key.bias"]
    Wv, bv = p[LAYER + "attention.value.weight"], p[LAYER + "attention.value.bias"]
    Wd, bd = p[LAYER + "attention.dense.weight"], p[LAYER + "attention.dense.bias"]
    g1, b1 = p[LAYER + "attention.LayerNorm.weight"], p[LAYER + "attention.LayerNorm.bias"]
    W1, c1 = p[LAYER + "ffn.weight"], p[LAYER + "ffn.bias"]
    W2, c2 = p[LAYER + "ffn_output.weight"], p[LAYER + "ffn_output.bias"]
    g2, b2 = p[LAYER + "full_layer_layer_norm.weight"], p[LAYER + "full_layer_layer_norm.bias"]
    scale = dtype(d ** -0.5)

    for _ in range(cfg.num_hidden_layers):  # the SAME weights every iteration (modeling_albert.py:274-282)
        q = _lin(x, Wq, bq).reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        k = _lin(x, Wk, bk).reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        v = _lin(x, Wv, bv).reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) * scale
        if bias is not None:
            s = s + bias
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        a, c_ln1 = layer_norm_fwd(x + _lin(ctx, Wd, bd), g1, b1, eps)
        u = _lin(a, W1, c1)
        gact = gelu_new(u)
        y, c_ln2 = layer_norm_fwd(_lin(gact, W2, c2) + a, g2, b2, eps)
        if keep:
            caches["layers"].append(dict(x=x, q=q, k=k, v=v, pr=pr, ctx=ctx, ln1=c_ln1, a=a, u=u,
                                         g=gact, ln2=c_ln2))
        x = y
    return x, caches


def heads_forward(cfg, P, hidden, dtype=np.float32):
    """model.py:13-18 / 26-30: phoneme_pred (and token_pred when the params are present)."""
    Wp = np.asarray(P["phoneme_predictor.weight"], dtype)
    bp = np.asarray(P["phoneme_predictor.bias"], dtype)
    out = [hidden @ Wp.T + bp]
    if "token_predictor.weight" in P:
        Wt = np.asarray(P["token_predictor.weight"], dtype)
        bt = np.asarray(P["token_predictor.bias"], dtype)
        out.append(hidden @ Wt.T + bt)
    return out


def model_forward(cfg, P, ids, attention_mask=None, dtype=np.float32):
    """PhonemeOnlyModel.forward / MultiTaskModel.forward → logits (tuple when a token head exists)."""
    h, caches = encoder_forward(cfg, P, ids, attention_mask, dtype)
    outs = heads_forward(cfg, P, h, dtype)
    return (outs[0] if len(outs) == 1 else tuple(outs)), h, caches


def log_softmax(z):
    z = z - z.max(-1, keepdims=True)
    return z - np.log(np.exp(z).sum(-1, keepdims=True))


def phoneme_loss(pred, labels, lengths, masked_indices):
    """calculate_phoneme_loss (train.py:107-131) with nn.CrossEntropyLoss() (mean reduction).

    Returns (loss, dpred) where dpred = d loss / d pred, zero outside the indexed rows.
    When no sample has a masked index the reference returns tensor(0.0): loss 0, dpred 0.
    """
    dpred = np.zeros_like(pred)
    nonempty = [b for b, idx in enumerate(masked_indices) if len(idx) > 0]
    count = len(nonempty)
    if count == 0:
        return pred.dtype.type(0.0), dpred
    total = pred.dtype.type(0.0)
    for b in nonempty:
        idx = np.asarray(masked_indices[b], dtype=np.int64)
        L = int(lengths[b])
        rows = pred[b, :L][idx]
        tgt = np.asarray(labels[b, :L])[idx]
        lsm = log_softmax(rows)
        n = len(idx)
        total = total + (-lsm[np.arange(n), tgt]).mean()
        g = np.exp(lsm)
        g[np.arange(n), tgt] -= 1.0
        g /= (n * count)
        np.add.at(dpred[b], idx, g)  # duplicate indices accumulate, as autograd's index backward
    return total / count, dpred


def token_loss(pred, token_ids, lengths):
    """Grapheme/token-head loss. The reference defines the head (model.py:11,16) and the 4-tuple batch
    that feeds it (dataloader.py:200-223) but no loss for it — train.py trains PhonemeOnlyModel only — so
    this follows upstream PL-BERT's training loop (yl4579/PL-BERT train.py, ``loss_vocab``): per-sample
    CrossEntropyLoss (mean) over the valid positions [:length], averaged over the B samples.
    PARITY UNPINNED by the reference for the loss formula; the logits it consumes are pinned (model.py:16),
    and tests/golden/*_dualloss.npz pins this restatement against torch autograd of exactly this formula
    applied to the reference's MultiTaskModel.

    Returns (loss, dpred): dpred = d loss / d pred, zero at padded positions.
    """
    B = pred.shape[0]
    dpred = np.zeros_like(pred)
    total = pred.dtype.type(0.0)
    for b in range(B):
        L = int(lengths[b])
        rows = pred[b, :L]
        tgt = np.asarray(token_ids[b, :L], dtype=np.int64)
        lsm = log_softmax(rows)
        total = total + (-lsm[np.arange(L), tgt]).mean()
        g = np.exp(lsm)
        g[np.arange(L), tgt] -= 1.0
        dpred[b, :L] = g / (L * B)
    return total / B, dpred


def encoder_backward(cfg, P, caches, dh, dtype=np.float32):
    """Gradient of everything under ``encoder.`` given d loss / d last_hidden_state."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
    ids = caches["ids"]
    B, S = ids.shape
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    scale = dtype(d ** -0.5)
    G = {}

    def acc(name, val):
        G[name] = G[name] + val if name in G else val

    Wq, Wk, Wv = (p[LAYER + "attention.query.weight"], p[LAYER + "attention.key.weight"],
                  p[LAYER + "attention.value.weight"])
    Wd = p[LAYER + "attention.dense.weight"]
    g1 = p[LAYER + "attention.LayerNorm.weight"]
    W1, W2 = p[LAYER + "ffn.weight"], p[LAYER + "ffn_output.weight"]
    g2 = p[LAYER + "full_layer_layer_norm.weight"]

    dy = dh
    for c in reversed(caches["layers"]):
        T = B * S
        dpre2, dg2, db2 = layer_norm_bwd(dy, c["ln2"], g2)
        acc(LAYER + "full_layer_layer_norm.weight", dg2)
        acc(LAYER + "full_layer_layer_norm.bias", db2)
        acc(LAYER + "ffn_output.weight", dpre2.reshape(T, H).T @ c["g"].reshape(T, -1))
        acc(LAYER + "ffn_output.bias", dpre2.reshape(T, H).sum(0))
        du = (dpre2 @ W2) * gelu_new_grad(c["u"])
        acc(LAYER + "ffn.weight", du.reshape(T, -1).T @ c["a"].reshape(T, H))
        acc(LAYER + "ffn.bias", du.reshape(T, -1).sum(0))
        da = du @ W1 + dpre2
        dpre1, dg1, db1 = layer_norm_bwd(da, c["ln1"], g1)
        acc(LAYER + "attention.LayerNorm.weight", dg1)
        acc(LAYER + "attention.LayerNorm.bias", db1)
        acc(LAYER + "attention.dense.weight", dpre1.reshape(T, H).T @ c["ctx"].reshape(T, H))
        acc(LAYER + "attention.dense.bias", dpre1.reshape(T, H).sum(0))
        dctx = (dpre1 @ Wd).reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        pr, q, k, v = c["pr"], c["q"], c["k"], c["v"]
        dv = pr.transpose(0, 1, 3, 2) @ dctx
        dpr = dctx @ v.transpose(0, 1, 3, 2)
        ds = pr * (dpr - (dpr * pr).sum(-1, keepdims=True))
        dq = (ds @ k) * scale
        dk = (ds.transpose(0, 1, 3, 2) @ q) * scale
        dq = dq.transpose(0, 2, 1, 3).reshape(T, H)
        dk = dk.transpose(0, 2, 1, 3).reshape(T, H)
        dv = dv.transpose(0, 2, 1, 3).reshape(T, H)
        x2 = c["x"].reshape(T, H)
        for nm, dz, W in (("query", dq, Wq), ("key", dk, Wk), ("value", dv, Wv)):
            acc(LAYER + f"attention.{nm}.weight", dz.T @ x2)
            acc(LAYER + f"attention.{nm}.bias", dz.sum(0))
        dy = (dq @ Wq + dk @ Wk + dv @ Wv).reshape(B, S, H) + dpre1

    T = B * S
    e2 = caches["e"].reshape(T, -1)
    dx0 = dy.reshape(T, H)
    G[ENC + "encoder.embedding_hidden_mapping_in.weight"] = dx0.T @ e2
    G[ENC + "encoder.embedding_hidden_mapping_in.bias"] = dx0.sum(0)
    de = (dx0 @ p[ENC + "encoder.embedding_hidden_mapping_in.weight"]).reshape(B, S, -1)
    dsum, dge, dbe = layer_norm_bwd(de, caches["eln"], p[ENC + "embeddings.LayerNorm.weight"])
    G[ENC + "embeddings.LayerNorm.weight"] = dge
    G[ENC + "embeddings.LayerNorm.bias"] = dbe
    dword = np.zeros_like(p[ENC + "embeddings.word_embeddings.weight"])
    np.add.at(dword, ids.reshape(-1), dsum.reshape(T, -1))
    # nn.Embedding(padding_idx=0): the pad row receives no gradient (modeling_albert.py:56)
    dword[0] = 0
    G[ENC + "embeddings.word_embeddings.weight"] = dword
    dpos = np.zeros_like(p[ENC + "embeddings.position_embeddings.weight"])
    dpos[:S] = dsum.sum(0)
    G[ENC + "embeddings.position_embeddings.weight"] = dpos
    dtyp = np.zeros_like(p[ENC + "embeddings.token_type_embeddings.weight"])
    dtyp[0] = dsum.sum((0, 1))
    G[ENC + "embeddings.token_type_embeddings.weight"] = dtyp
    return G


def loss_and_grads(cfg, P, masked_ids, labels, lengths, masked_indices, dtype=np.float32, token_ids=None):
    """process_batch (train.py:381-390) + backward: (loss, logits, grads-by-name).

    Parameters that receive no gradient in the reference (the pooler) are absent from ``grads``.
    With ``token_ids`` (the 4-tuple batch, dataloader.py:200-223) and a token head in P the loss is
    phoneme_loss + token_loss (dual-head training) and the second return value is (phoneme_pred, token_pred).
    """
    am = attention_mask_from_lengths(lengths)
    h, caches = encoder_forward(cfg, P, masked_ids, am, dtype)
    Wp = np.asarray(P["phoneme_predictor.weight"], dtype)
    bp = np.asarray(P["phoneme_predictor.bias"], dtype)
    pred = h @ Wp.T + bp
    loss, dpred = phoneme_loss(pred, np.asarray(labels), lengths, masked_indices)
    B, S, H = h.shape
    G = {}
    G["phoneme_predictor.weight"] = dpred.reshape(B * S, -1).T @ h.reshape(B * S, H)
    G["phoneme_predictor.bias"] = dpred.reshape(B * S, -1).sum(0)
    dh = dpred @ Wp
    if token_ids is not None:
        Wt = np.asarray(P["token_predictor.weight"], dtype)
        bt = np.asarray(P["token_predictor.bias"], dtype)
        tpred = h @ Wt.T + bt
        tl, dtp = token_loss(tpred, np.asarray(token_ids), lengths)
        G["token_predictor.weight"] = dtp.reshape(B * S, -1).T @ h.reshape(B * S, H)
        G["token_predictor.bias"] = dtp.reshape(B * S, -1).sum(0)
        dh = dh + dtp @ Wt
        G.update(encoder_backward(cfg, P, caches, dh, dtype))
        return loss + tl, (pred, tpred), G
    G.update(encoder_backward(cfg, P, caches, dh, dtype))
    return loss, pred, G


class AdamW:
    """torch.optim.AdamW(params, lr) with torch defaults (train.py:272): betas (0.9, 0.999),
    eps 1e-8, weight_decay 0.01 on every parameter that has a gradient; parameters whose grad is
    None (the pooler) are skipped entirely."""

    def __init__(self, lr=7e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, P, G):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        for name, g in G.items():
            p = P[name]
            g = g.astype(p.dtype)
            if name not in self.m:
                self.m[name] = np.zeros_like(p)
                self.v[name] = np.zeros_like(p)
            p *= (1.0 - self.lr * self.wd)
            m = self.m[name]
            v = self.v[name]
            m *= b1
            m += (1.0 - b1) * g
            v *= b2
            v += (1.0 - b2) * g * g
            denom = np.sqrt(v) / math.sqrt(bc2) + self.eps
            p -= (self.lr / bc1) * (m / denom)


def train_step(cfg, P, opt, masked_ids, labels, lengths, masked_indices, dtype=np.float32, token_ids=None):
    """One iteration of the loop body train.py:350-357. Mutates P in place; returns the loss."""
    loss, _, G = loss_and_grads(cfg, P, masked_ids, labels, lengths, masked_indices, dtype, token_ids)
    opt.step(P, G)
    return loss
