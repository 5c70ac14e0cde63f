"""CPU restatement of the fp8 training call (BASELINE.json configs[4]).  TEST INFRASTRUCTURE ONLY.

Same rule as ``albert_np.py``: only ``tests/`` may import this file; the product path never does.

The reference has no fp8 path (it trains under fp16 autocast, configs/config.yml:15), so nothing here is pinned by the
reference beyond what ``albert_np.py`` already pins: with ``amax=None`` this module computes EXACTLY the arithmetic of
``albert_np.loss_and_grads`` (tests/test_oracle_fp8.py asserts equality) and records the per-site maxima; with maxima
given it repeats the call with the operands of the layer's large GEMMs rounded to OCP fp8 at the sites, in the formats
and under the scales the HIP path uses (plbert_amd/csrc/engine.cpp "fp8 mode"):

  site   tensor (per application l)                 format   consumers
  X      layer input x_l                            e4m3     QKV GEMM, dW_qkv
  C      attention context                          e4m3     dense GEMM, dW_dense
  A      LayerNorm-1 output a                       e4m3     FFN-up GEMM, dW_ffn
  G      gelu_new(u)                                e4m3     FFN-output GEMM, dW_ffn_output
  DP     d pre-LayerNorm-2  (dpre2)                 e5m2     dU GEMM, dW_ffn_output
  DU     dU = (dpre2 W2) * gelu'(u)                 e5m2     dA GEMM, dW_ffn
  DP1    d pre-LayerNorm-1  (dpre1)                 e5m2     dCtx GEMM, dW_dense
  DQ     d[Q|K|V]                                   e5m2     dX GEMM, dW_qkv
  weights Wqkv (one tensor), Wd, W1, W2             e4m3     per-tensor scale 448 / max|W|

One scale per site, shared by its L applications: 448 / max_l amax for the activations, 28672 / max_l amax for the
gradients (half of e5m2's range as headroom), amax taken from the PREVIOUS call (delayed scaling; the first call after
``set_fp8`` computes in bf16 and only records).  Residual additions, LayerNorm, attention, biases, bias gradients, the
head and the loss stay un-rounded (bf16 on the device, ``dtype`` here).  ``tn8=False`` restates PLBERT_FP8_TN=0 / the
row-minimum fall-back: weight gradients from the un-rounded tensors.

``bf16=True`` additionally rounds to bfloat16 wherever the device STORES a tensor in bfloat16 (every GEMM / LayerNorm /
attention output, the bf16 weight copies, the gelu-derivative stash, the probabilities and dS fed to the attention
products) — csrc/engine.cpp run_encoder / backward; the fused LayerNorm epilogues round their input "as a bf16 store
would leave it" (gemm_nt_pipeline.h), so fused and unfused calls share these semantics.  The 1-byte images are taken
from the bf16 values (attn_common.h store_transposed, gemm_nt_pipeline.h "the values as stored"), so with the stores
restated the images — and with them the rounding decisions of every later site — coincide with the device's almost
everywhere; without it a ~1 % of the elements per site fall on the other side of an fp8 rounding boundary (bf16 ulp /
fp8 ulp = 2^-5 per perturbed element) and the difference compounds through the sites.  Not restated: the forward
attention kernel's running maximum (P is rounded under the maximum of the key tiles seen so far), accumulation order,
the device's exp2/rcp approximations.

What this buys: the HIP fp8 step is compared with its own bf16 step at ~0.1 relative L2 (3 mantissa bits); against
THIS restatement (bf16=True) only the un-restated details above remain, so a wrong scale, format, site or
dequantisation factor shows up as a many-fold larger distance (tests/test_gpu_fp8.py).
"""
from __future__ import annotations

import numpy as np

from . import albert_np as ref
from .albert_np import ENC, LAYER

E4M3 = dict(mant=3, emin=-6, fmax=448.0)       # OCP e4m3fn: bias 7, subnormal step 2^-9, no infinities
E5M2 = dict(mant=2, emin=-14, fmax=57344.0)    # OCP e5m2: bias 15, subnormal step 2^-16
ACT_SITES = ("X", "C", "A", "G")
GRAD_SITES = ("DP", "DU", "DP1", "DQ")
GRAD_TARGET = 28672.0


def round_fp8(x, fmt):
    """Round to the nearest value of the format (ties to even), saturating at +-fmax — v_cvt_pk_{fp8,bf8}_f32 after the
    clamp of csrc/common.h pack_fp8x4."""
    x = np.clip(np.asarray(x, np.float64), -fmt["fmax"], fmt["fmax"])
    _, ex = np.frexp(x)                                   # |x| = m 2^ex, m in [0.5, 1)
    e = np.maximum(ex - 1, fmt["emin"])                   # subnormals share the smallest normal's spacing
    step = np.ldexp(1.0, e - fmt["mant"])
    return np.rint(x / step) * step


def round_bf16(x):
    """Round to bfloat16 (nearest, ties to even) — csrc/common.h pack_bf2."""
    x32 = np.ascontiguousarray(x, np.float32)
    u = x32.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32).astype(np.asarray(x).dtype)


def fake_quant(x, amax, fmt, target=None):
    """dequantise(quantise(x)): x -> round(x * s) / s with s = target / amax; the scale and the scaled value in fp32 as
    on the device (rowops.hip fp8_scales_kernel: scale = fmax / amax, deq = amax / fmax)."""
    if not (amax > 0 and np.isfinite(amax)):
        return x
    t = np.float32(fmt["fmax"] if target is None else target)
    s = t / np.float32(amax)
    deq = np.float32(amax) / t
    return (round_fp8(np.asarray(x, np.float32) * s, fmt) * np.float64(deq)).astype(x.dtype)


class _Sites:
    """Identity + recording when ``amax`` is None, else fake-quantisation under the given maxima (and recording)."""

    def __init__(self, amax, bf16):
        self.use = amax
        self.bf16 = bf16
        self.seen = {}

    def __call__(self, site, x):
        self.seen[site] = max(self.seen.get(site, 0.0), float(np.abs(x).max()))
        if self.use is None:
            return x
        if site in GRAD_SITES:
            return fake_quant(x, self.use[site], E5M2, GRAD_TARGET)
        return fake_quant(x, self.use[site], E4M3)

    def weights(self, w):
        """(copy the forward GEMM reads, copy the backward GEMM reads) of one weight tensor.  fp8 call: the forward copy
        is quantised from the fp32 master, the backward (transposed) copy from its bf16 transpose — each under its own
        maximum (engine.cpp fp8_quantize_weights).  bf16 call: the bf16 copy both ways."""
        w16 = round_bf16(w) if self.bf16 else w
        if self.use is None:
            return w16, w16
        return fake_quant(w, float(np.abs(w).max()), E4M3), fake_quant(w16, float(np.abs(w16).max()), E4M3)


def loss_and_grads_fp8(cfg, P, masked_ids, labels, lengths, masked_indices, amax=None, tn8=True, bf16=False,
                       dtype=np.float64, prune_last=False):
    """One loss call: (loss, phoneme_pred, grads-by-name, maxima recorded per site).

    prune_last: restates the device's evaluation of the LAST application's post-attention part on the masked rows only
    (csrc/engine.cpp last_application_fwd_pruned; a phoneme-only call with at most half of the positions masked). That part
    runs in bf16 even inside an fp8 call — dense, FFN and their dX GEMMs on the bf16 weight copies and un-rounded operands,
    the pre-activation u stored in bf16 with gelu / gelu' taken from the stored value — while the stacked weight-gradient
    GEMMs still read 1-byte images of its (compact) rows. Rows without a masked position carry an exactly zero gradient
    through that part, so evaluating every row here gives the device's sums.

    amax=None: the calibration call — plain arithmetic (== albert_np.loss_and_grads when bf16=False).  amax = the dict a
    previous call returned: the fp8 call.  bf16=True: bfloat16 stores as on the device (module docstring).  Follows
    albert_np.encoder_forward / encoder_backward line for line otherwise."""
    Q = _Sites(amax, bf16)
    r16 = round_bf16 if bf16 else (lambda t: t)
    p = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
    ids = np.asarray(masked_ids)
    B, S = ids.shape
    T = B * S
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    eps = cfg.layer_norm_eps
    scale = dtype(d ** -0.5)
    am = ref.attention_mask_from_lengths(lengths)

    e_sum = (p[ENC + "embeddings.word_embeddings.weight"][ids]
             + p[ENC + "embeddings.token_type_embeddings.weight"][0][None, None, :]
             + p[ENC + "embeddings.position_embeddings.weight"][:S][None, :, :])
    e, c_eln = ref.layer_norm_fwd(e_sum, p[ENC + "embeddings.LayerNorm.weight"], p[ENC + "embeddings.LayerNorm.bias"], eps)
    e = r16(e)
    Win, b_in = r16(p[ENC + "encoder.embedding_hidden_mapping_in.weight"]), p[ENC + "encoder.embedding_hidden_mapping_in.bias"]
    x = r16(e @ Win.T + b_in)
    bias = None
    if not (am != 0).all():
        bias = np.where(am[:, None, None, :] != 0, 0.0, np.finfo(dtype).min).astype(dtype)

    Wq, Wk, Wv = (p[LAYER + f"attention.{n}.weight"] for n in ("query", "key", "value"))
    bqkv = np.concatenate([p[LAYER + f"attention.{n}.bias"] for n in ("query", "key", "value")])
    Wd, bd = p[LAYER + "attention.dense.weight"], p[LAYER + "attention.dense.bias"]
    g1, b1 = p[LAYER + "attention.LayerNorm.weight"], p[LAYER + "attention.LayerNorm.bias"]
    W1, c1 = p[LAYER + "ffn.weight"], p[LAYER + "ffn.bias"]
    W2, c2 = p[LAYER + "ffn_output.weight"], p[LAYER + "ffn_output.bias"]
    g2, b2 = p[LAYER + "full_layer_layer_norm.weight"], p[LAYER + "full_layer_layer_norm.bias"]
    # Q, K and V are ONE tensor on the device ([3H, H], one scale)
    Wqkv_f, Wqkv_b = Q.weights(np.concatenate([Wq, Wk, Wv], 0))
    Wd_f, Wd_b = Q.weights(Wd)
    W1_f, W1_b = Q.weights(W1)
    W2_f, W2_b = Q.weights(W2)

    def heads(t):  # [T, 3H] -> q, k, v as [B, nh, S, d]
        return (t[:, i * H:(i + 1) * H].reshape(B, S, nh, d).transpose(0, 2, 1, 3) for i in range(3))

    layers = []
    L = cfg.num_hidden_layers
    w16 = {k: (round_bf16(v) if bf16 else v) for k, v in (("d", Wd), ("1", W1), ("2", W2))}
    for li in range(L):
        cpt = prune_last and li == L - 1                         # the compact part: bf16 launches, also inside an fp8 call
        x8 = Q("X", x)
        qkv = r16(x8.reshape(T, H) @ Wqkv_f.T + bqkv)
        q, k, v = heads(qkv)
        s = (q @ k.transpose(0, 1, 3, 2)) * scale
        if bias is not None:
            s = s + bias
        s = s - s.max(-1, keepdims=True)
        pe = np.exp(s)
        lsum = pe.sum(-1, keepdims=True)
        pr = pe / lsum
        ctx = r16(((r16(pe) @ v) / lsum).transpose(0, 2, 1, 3).reshape(B, S, H))   # P rounded for the second product only
        c8 = Q("C", ctx)
        pre1 = r16(x + ((ctx @ w16["d"].T if cpt else c8 @ Wd_f.T) + bd))         # the residual is the un-rounded (bf16) x
        a, ln1 = ref.layer_norm_fwd(pre1, g1, b1, eps)
        a = r16(a)
        a8 = Q("A", a)
        if cpt:   # act 1 / act 2 launches: u is STORED (bf16); gelu from the stored value, gelu' evaluated in the backward epilogue
            u = r16(a @ w16["1"].T + c1)
            gact, dgelu = r16(ref.gelu_new(u)), ref.gelu_new_grad(u)
        else:
            u = a8 @ W1_f.T + c1                                                    # never stored: gelu and gelu' from the fp32 value
            gact, dgelu = r16(ref.gelu_new(u)), r16(ref.gelu_new_grad(u))
        g8 = Q("G", gact)
        pre2 = r16(((gact @ w16["2"].T if cpt else g8 @ W2_f.T) + c2) + a)
        y, ln2 = ref.layer_norm_fwd(pre2, g2, b2, eps)
        y = r16(y)
        layers.append(dict(x=x, x8=x8, q=q, k=k, v=v, pr=pr, ctx=ctx, c8=c8, ln1=ln1, a=a, a8=a8, dgelu=dgelu, g=gact, g8=g8,
                           ln2=ln2, cpt=cpt))
        x = y
    h = x

    Wp, bp = r16(p["phoneme_predictor.weight"]), p["phoneme_predictor.bias"]
    pred = h @ Wp.T + bp
    loss, dpred = ref.phoneme_loss(pred, np.asarray(labels), lengths, masked_indices)
    dpred = r16(dpred)
    G = {"phoneme_predictor.weight": dpred.reshape(T, -1).T @ h.reshape(T, H),
         "phoneme_predictor.bias": dpred.reshape(T, -1).sum(0)}
    dy = r16(dpred @ Wp)

    def acc(name, val):
        G[name] = G[name] + val if name in G else val

    for c in reversed(layers):
        dpre2, dg2, db2 = ref.layer_norm_bwd(dy, c["ln2"], g2)
        acc(LAYER + "ffn_output.bias", dpre2.reshape(T, H).sum(0))
        dpre2 = r16(dpre2)
        dp8 = Q("DP", dpre2)
        acc(LAYER + "full_layer_layer_norm.weight", dg2)
        acc(LAYER + "full_layer_layer_norm.bias", db2)
        du = r16(((dpre2 @ w16["2"]) if c["cpt"] else (dp8 @ W2_b)) * c["dgelu"])
        du8 = Q("DU", du)
        acc(LAYER + "ffn.bias", du.reshape(T, -1).sum(0))
        da = r16(((du @ w16["1"]) if c["cpt"] else (du8 @ W1_b)) + dpre2)
        dpre1, dg1, db1 = ref.layer_norm_bwd(da, c["ln1"], g1)
        acc(LAYER + "attention.dense.bias", dpre1.reshape(T, H).sum(0))
        dpre1 = r16(dpre1)
        dp18 = Q("DP1", dpre1)
        acc(LAYER + "attention.LayerNorm.weight", dg1)
        acc(LAYER + "attention.LayerNorm.bias", db1)
        dctx = r16((dpre1 @ w16["d"]) if c["cpt"] else (dp18 @ Wd_b))
        dO = dctx.reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        O = c["ctx"].reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        pr, q, k, v = c["pr"], c["q"], c["k"], c["v"]
        dv = r16(pr).transpose(0, 1, 3, 2) @ dO
        dpr = dO @ v.transpose(0, 1, 3, 2)
        # delta = rowsum(dO * O) from the stored context (== rowsum(dP * P) in exact arithmetic: albert_np's form)
        delta = (dO * O).sum(-1, keepdims=True) if bf16 else (dpr * pr).sum(-1, keepdims=True)
        ds = r16(pr * (dpr - delta))
        dq = (ds @ k) * scale
        dk = (ds.transpose(0, 1, 3, 2) @ q) * scale
        dqkv = r16(np.concatenate([t.transpose(0, 2, 1, 3).reshape(T, H) for t in (dq, dk, dv)], 1))   # [T, 3H]: one site
        dq8 = Q("DQ", dqkv)
        acc("_qkv_bias", dqkv.sum(0))
        # weight gradients: products of the two IMAGES (tn8) or of the stored tensors
        wa = (dq8, c["x8"], du8, c["a8"], dp8, c["g8"], dp18, c["c8"]) if tn8 else \
             (dqkv, c["x"], du, c["a"], dpre2, c["g"], dpre1, c["ctx"])
        acc("_qkv_weight", wa[0].T @ wa[1].reshape(T, H))
        acc(LAYER + "ffn.weight", wa[2].reshape(T, -1).T @ wa[3].reshape(T, H))
        acc(LAYER + "ffn_output.weight", wa[4].reshape(T, H).T @ wa[5].reshape(T, -1))
        acc(LAYER + "attention.dense.weight", wa[6].reshape(T, H).T @ wa[7].reshape(T, H))
        dy = r16((dq8 @ Wqkv_b).reshape(B, S, H) + dpre1)
    for i, n in enumerate(("query", "key", "value")):
        G[LAYER + f"attention.{n}.weight"] = G["_qkv_weight"][i * H:(i + 1) * H]
        G[LAYER + f"attention.{n}.bias"] = G["_qkv_bias"][i * H:(i + 1) * H]
    del G["_qkv_weight"], G["_qkv_bias"]

    # below the layers: bf16 on the device, plain here (albert_np.encoder_backward's tail)
    e2 = e.reshape(T, -1)
    dx0 = dy.reshape(T, H)
    G[ENC + "encoder.embedding_hidden_mapping_in.weight"] = dx0.T @ e2
    G[ENC + "encoder.embedding_hidden_mapping_in.bias"] = dx0.sum(0)
    de = r16(dx0 @ Win).reshape(B, S, -1)
    dsum, dge, dbe = ref.layer_norm_bwd(de, c_eln, p[ENC + "embeddings.LayerNorm.weight"])
    G[ENC + "embeddings.LayerNorm.weight"] = dge
    G[ENC + "embeddings.LayerNorm.bias"] = dbe
    dword = np.zeros_like(p[ENC + "embeddings.word_embeddings.weight"])
    np.add.at(dword, ids.reshape(-1), dsum.reshape(T, -1))
    dword[0] = 0
    G[ENC + "embeddings.word_embeddings.weight"] = dword
    dpos = np.zeros_like(p[ENC + "embeddings.position_embeddings.weight"])
    dpos[:S] = dsum.sum(0)
    G[ENC + "embeddings.position_embeddings.weight"] = dpos
    dtyp = np.zeros_like(p[ENC + "embeddings.token_type_embeddings.weight"])
    dtyp[0] = dsum.sum((0, 1))
    G[ENC + "embeddings.token_type_embeddings.weight"] = dtyp
    return loss, pred, G, dict(Q.seen)
