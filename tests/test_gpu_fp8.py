"""fp8 path (-m gpu; BASELINE.json configs[4]: "fp8 (e4m3) MFMA path for QKV/FFN GEMMs").

The reference has NO fp8 path (it trains under fp16 autocast, configs/config.yml:15), so there is no oracle: parity is
against this repository's own bf16 path, as SURVEY.md §7 and BASELINE.md §4 prescribe.
 * kernel level: the fp8 pipeline GEMM against fp32 torch arithmetic on the SAME quantised operands — the only
   difference left is accumulation order, so the tolerance is tight (1e-3 of the output scale after bf16 rounding);
 * step level: loss and gradients of the fp8 step against the bf16 step on identical weights and batch — tolerances
   2e-2 relative on the loss (fp8 carries 3 mantissa bits; the north-star 1e-3 is the bf16-vs-reference bar, not
   this one), per-tensor gradient relative L2 stated at each assertion; then both trainers run 12 steps and must
   descend together.
tests/golden/small_h128 cannot exercise the path (hidden 128 has no pipeline tile; plb_set_fp8 refuses it), so the
reference-captured fixture used is real_s128_b8 (768/12) plus a bench-shaped synthetic batch."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
from gpu_util import rel_l2, stream
import plbert_amd
from plbert_amd import _lib
from plbert_amd.engine import HipEngine
from plbert_amd.train import PLBertTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _q(x, scale, bf8):
    dt, mx = (torch.float8_e5m2, 57344.0) if bf8 else (torch.float8_e4m3fn, 448.0)
    q = (x.float() * scale).clamp(-mx, mx).to(dt)
    return q, q.float() / scale


@pytest.mark.parametrize("M,N,K,act,bf8", [(256, 768, 768, 0, 0), (384, 2304, 768, 0, 0), (256, 768, 2048, 0, 1),
                                           (256, 2048, 768, 1, 0), (256, 2048, 768, 2, 1), (128, 1024, 1024, 0, 0)])
def test_fp8_gemm_against_fp32_on_the_same_quantised_operands(M, N, K, act, bf8):
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(M + N + K + act)
    A = torch.randn(M, K, device=DEV, generator=g) * (0.01 if bf8 else 1.0)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.05
    sa, sw = (57344.0 if bf8 else 448.0) / float(A.abs().max()), 448.0 / float(W.abs().max())
    A8, Ad = _q(A, sa, bf8)
    W8, Wd = _q(W, sw, False)
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    res = (torch.randn(M, N, device=DEV, generator=g)).to(torch.bfloat16)
    aux = (torch.randn(M, N, device=DEV, generator=g)).to(torch.bfloat16)
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    C2 = torch.zeros_like(Cb)
    C8 = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
    deq = torch.tensor([1.0 / sa, 1.0 / sw], device=DEV)
    qs = torch.tensor([3.0], device=DEV)
    amax = torch.zeros(64 * 16, device=DEV)  # one site: 64 slots on separate 64-byte lines
    colp = torch.zeros(2 * (M // 128), N, device=DEV)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb = A8.view(torch.uint8).data_ptr(), K, W8.view(torch.uint8).data_ptr(), K
    p.M, p.N, p.K, p.Mstore = M, N, K, M - 3
    p.C, p.ldc, p.C2, p.ldc2 = Cb.data_ptr(), N, C2.data_ptr(), N
    p.deq_a, p.deq_b = deq.data_ptr(), deq.data_ptr() + 4
    if act == 0:
        p.bias, p.res, p.ldr = bias.data_ptr(), res.data_ptr(), N
    if act == 1:
        p.bias = bias.data_ptr()
    if act == 2:
        p.aux, p.ldaux, p.colpart = aux.data_ptr(), N, colp.data_ptr()
    if act:
        p.C8, p.ldc8, p.q_scale, p.q_amax, p.c8_bf8 = C8.data_ptr(), N, qs.data_ptr(), amax.data_ptr(), int(act == 2)
    assert L.plb_launch_gemm_nt_fp8(C.byref(p), act, bf8, stream()) == 0
    torch.cuda.synchronize()
    ref = Ad.double() @ Wd.double().T
    rows = slice(0, M - 3)
    if act == 0:
        want = ref + bias.double() + res.double()
        assert rel_l2(Cb[rows].float(), want[rows].float()) < 4e-3          # bf16 rounding of the output
        assert (Cb[M - 3:] == 0).all()                                       # rows >= Mstore are not stored
    elif act == 1:
        u = (ref + bias.double()).float().to(torch.bfloat16)
        assert rel_l2(Cb[rows].float(), u[rows].float()) < 4e-3
        gl = torch.nn.functional.gelu(Cb.float(), approximate="tanh")
        assert rel_l2(C2[rows].float(), gl[rows]) < 4e-3
        q8 = C8.view(torch.float8_e4m3fn).float() / 3.0                     # the fp8 image of gelu as stored
        assert rel_l2(q8[rows], C2[rows].float()) < 4e-2                     # 3 mantissa bits
        assert abs(float(amax.max()) - float(C2[rows].float().abs().max())) < 1e-6
    else:
        x = aux.float()
        k = 0.7978845608028654
        t = torch.tanh(k * (x + 0.044715 * x ** 3))
        dgelu = 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * k * (1 + 3 * 0.044715 * x * x)
        want = ref.float() * dgelu
        assert rel_l2(Cb[rows].float(), want[rows]) < 6e-3
        q8 = C8.view(torch.float8_e5m2).float() / 3.0
        assert rel_l2(q8[rows], Cb[rows].float()) < 8e-2                     # 2 mantissa bits
        assert rel_l2(colp.sum(0), Cb[rows].float().sum(0)) < 1e-4           # bias-gradient partials of the stored values


def _pair(cfg, B, S, sd=None, seed=0):
    trs = []
    for fp8 in (False, True):
        tr = PLBertTrainer(cfg, 188, max_batch=B, max_seq=S, lr=1e-4, seed=seed, state_dict=sd)
        if fp8:
            tr.engine.set_fp8(True)
        trs.append(tr)
    return trs


def test_fp8_step_against_own_bf16_step_real_model_fixture():
    """real_s128_b8: the reference-captured 768/12 fixture (weights, batch). Call 1 calibrates (bf16 arithmetic: must
    equal the bf16 engine bit for bit), call 2 runs the fp8 GEMMs."""
    g = load_golden("real_s128_b8")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    idx = [list(map(int, x)) for x in g["index"]]
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    args = (g["masked"], g["labels"], g["lengths"].astype(np.int32), off, flat, int(off[-1]))
    ref = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S)
    ref.load_state_dict(sd)
    l_ref = float(ref.loss_fwd_bwd(*args).item())
    eng = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    eng.set_fp8(True)
    assert eng.fp8_state() == (True, False)
    l_cal = float(eng.loss_fwd_bwd(*args).item())
    assert l_cal == l_ref and torch.equal(eng.grads, ref.grads)           # calibration call = the bf16 path
    assert eng.fp8_state() == (True, True)
    l_f8 = float(eng.loss_fwd_bwd(*args).item())
    torch.cuda.synchronize()
    assert l_f8 != l_ref                                                     # the fp8 GEMMs really ran
    assert abs(l_f8 - l_ref) / l_ref < 2e-2, (l_f8, l_ref)
    assert abs(l_f8 - float(g["loss"])) / float(g["loss"]) < 2e-2           # and stays near the reference's loss
    lp = "encoder.encoder.albert_layer_groups.0.albert_layers.0."
    for k, tol in ((lp + "ffn.weight", 0.2), (lp + "ffn_output.weight", 0.2), (lp + "attention.query.weight", 0.25),
                   (lp + "attention.dense.weight", 0.2), ("phoneme_predictor.weight", 0.1),
                   ("encoder.embeddings.word_embeddings.weight", 0.25)):
        r = rel_l2(eng.view(k, of=eng.grads), ref.view(k, of=ref.grads))
        assert r < tol, (k, r)
    # validation-only call and switching the mode off again
    l_val = float(eng.loss_fwd(*args).item())
    assert abs(l_val - l_ref) / l_ref < 2e-2
    eng.set_fp8(False)
    assert float(eng.loss_fwd_bwd(*args).item()) == l_ref


def test_fp8_training_descends_with_the_bf16_run():
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                  max_position_embeddings=512, num_hidden_layers=4)
    B, S = 8, 512
    labels, masked, lens, idx = plbert_amd.synthetic_batch(B, S, seed=5)
    lens = [512, 512, 400, 512, 300, 512, 512, 77]
    idx = [[i for i in ix if i < n] or [0] for ix, n in zip(idx, lens)]
    for b, n in enumerate(lens):
        labels[b, n:] = 0
        masked[b, n:] = 0
    tb, tf = _pair(cfg, B, S)
    bb, bf = tb.stage_batch(labels, masked, lens, idx), tf.stage_batch(labels, masked, lens, idx)
    lb = [float(tb.step(bb).item()) for _ in range(12)]
    lf = [float(tf.step(bf).item()) for _ in range(12)]
    torch.cuda.synchronize()
    assert lf[0] == lb[0]                                                    # step 1 calibrates in bf16
    assert all(np.isfinite(lf)) and lf[-1] < lf[0] - 0.05                    # it learns
    assert max(abs(a - b) / b for a, b in zip(lf, lb)) < 3e-2, (lf, lb)      # and tracks the bf16 run
    assert not torch.isnan(tf.engine.params).any()
