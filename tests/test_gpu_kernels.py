"""Kernel-level parity (-m gpu): each HIP kernel, launched through the C ABI, against a plain
PyTorch fp32 reference of the same op on the same bf16-rounded inputs."""
import ctypes as C
import math

import pytest
import torch

from plbert_amd import _lib
from gpu_util import attn_args, gemm_nt, gemm_tn, rel_l2, stream, torch_attention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gelu_new(x):
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def randbf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


# (the 128x128 kernel; grids of at most 256 workgroups with K >= 384 take its four-stage "deep" K loop: K-tile counts of 6, 7,
#  12, 32 and 64 — every tail length of the counted waits)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (384, 768, 2048), (128, 188, 768), (128, 128, 384),
                                   (256, 128, 448), (2048, 768, 2048), (1024, 1024, 4096), (128, 256, 320)])
def test_gemm_nt_bias_residual(M, N, K):
    A, Bw = randbf(M, K, seed=1), randbf((N + 127) // 128 * 128, K, scale=0.05, seed=2)
    bias = torch.randn(N, device=DEV)
    res = randbf(M, N, seed=3)
    out, _ = gemm_nt(A, Bw, N, bias=bias, res=res)
    ref = A.float() @ Bw[:N].float().T + bias + res.float()
    assert (out.float() - ref).abs().max() <= 1e-2 * ref.abs().max() + 1e-3
    assert rel_l2(out.float(), ref) < 4e-3  # bf16 output rounding
    outf, _ = gemm_nt(A, Bw, N, bias=bias, out_f32=True, Mstore=M - 5)
    reff = A.float() @ Bw[:N].float().T + bias
    assert rel_l2(outf[: M - 5], reff[: M - 5]) < 1e-5
    assert (outf[M - 5:] == 0).all()  # rows past Mstore untouched


@pytest.mark.parametrize("tile,M,N,K", [(256, 256, 256, 64), (256, 512, 768, 768), (256, 256, 512, 128), (256, 512, 256, 192),
                                        (384, 128, 384, 64), (384, 384, 768, 768), (384, 256, 2304, 128), (384, 128, 384, 2048),
                                        (1256, 128, 256, 64), (1256, 384, 1024, 128), (1256, 256, 512, 192), (1256, 128, 256, 1024)])
@pytest.mark.parametrize("prefetch", [1, 0])
def test_gemm_nt_big_tiles(tile, M, N, K, prefetch):
    """256x256 / 128x384 / 128x256 multi-phase kernels: 1, 2, 3 and many K-tiles (prologue, steady state,
    drain), in both K-loop forms (fragment-prefetching = default, staggered)."""
    L = _lib.lib()
    L.plb_set_gemm_nt_prefetch(prefetch)
    A, Bw = randbf(M, K, seed=21), randbf(N, K, scale=0.05, seed=22)
    bias = torch.randn(N, device=DEV)
    res = randbf(M, N, seed=23)
    ref = A.float() @ Bw.float().T + bias + res.float()
    try:
        L.plb_set_gemm_nt_tile(tile)
        out, _ = gemm_nt(A, Bw, N, bias=bias, res=res)
        outf, _ = gemm_nt(A, Bw, N, bias=bias, out_f32=True)
        u, g = gemm_nt(A, Bw, N, bias=bias, act=1)
    finally:
        L.plb_set_gemm_nt_tile(0)
        L.plb_set_gemm_nt_prefetch(-1)
    assert rel_l2(out.float(), ref) < 4e-3
    assert (out.float() - ref).abs().max() <= 1e-2 * ref.abs().max() + 1e-3
    assert rel_l2(outf, A.float() @ Bw.float().T + bias) < 1e-5
    assert rel_l2(g.float(), gelu_new(u.float())) < 5e-3


def test_big_gemm_race_screen():
    """The multi-phase kernels hand LDS slots between DMA and readers by counted waits and barriers: a
    mistake there shows as rare wrong tiles. Repeat launches must be bitwise identical and correct."""
    L = _lib.lib()
    for tile, (M, N, K) in ((256, (2048, 768, 2304)), (384, (2048, 2304, 768)), (256, (4096, 2048, 768)),
                            (1256, (2048, 1024, 1024))):
        A, Bw = randbf(M, K, seed=41), randbf(N, K, scale=0.05, seed=42)
        ref = A.float() @ Bw.float().T
        try:
            L.plb_set_gemm_nt_tile(tile)
            first, _ = gemm_nt(A, Bw, N)
            assert rel_l2(first.float(), ref) < 4e-3
            for _ in range(25):
                again, _ = gemm_nt(A, Bw, N)
                assert torch.equal(again, first)
        finally:
            L.plb_set_gemm_nt_tile(0)
    A, Bm = randbf(8192, 768, seed=43), randbf(8192, 512, seed=44)
    first = gemm_tn(A, Bm, 768, 8, 1024, big=True)
    assert rel_l2(first, A.float().T @ Bm.float()) < 1e-5
    for _ in range(25):
        assert torch.equal(gemm_tn(A, Bm, 768, 8, 1024, big=True), first)


def test_gemm_nt_gelu_epilogues():
    M, N, K = 256, 256, 128
    A, Bw = randbf(M, K, seed=4), randbf(N, K, scale=0.2, seed=5)
    bias = torch.randn(N, device=DEV) * 0.1
    u, g = gemm_nt(A, Bw, N, bias=bias, act=1)
    uref = A.float() @ Bw.float().T + bias
    assert rel_l2(u.float(), uref) < 4e-3
    assert (g.float() - gelu_new(u.float())).abs().max() < 2e-2 * max(1.0, float(u.float().abs().max()))
    assert rel_l2(g.float(), gelu_new(u.float())) < 5e-3
    # backward epilogue: (A·B^T) * gelu'(aux) + nothing
    aux = randbf(M, N, scale=1.5, seed=6)
    du, _ = gemm_nt(A, Bw, N, aux=aux, act=2)
    x = aux.float().clone().requires_grad_(True)
    gelu_new(x).sum().backward()
    ref = (A.float() @ Bw.float().T) * x.grad
    assert rel_l2(du.float(), ref) < 5e-3


@pytest.mark.parametrize("Mtot,Ncols,N,K,splits,rps", [(256, 128, 128, 128, 1, 256), (1024, 256, 188, 192, 3, 384),
                                                       (2048, 384, 384, 64, 4, 512), (640, 2304, 2304, 768, 2, 320)])
def test_gemm_tn(Mtot, Ncols, N, K, splits, rps):
    A, Bm = randbf(Mtot, Ncols, seed=7), randbf(Mtot, K, seed=8)
    out = gemm_tn(A, Bm, N, splits, rps)
    ref = A.float().T[:N] @ Bm.float()
    assert rel_l2(out, ref) < 1e-5


@pytest.mark.parametrize("Mtot,N,K,splits,rps", [(64, 256, 256, 1, 64), (128, 256, 512, 1, 128), (192, 512, 256, 1, 192),
                                                 (1024, 768, 256, 3, 384), (2048, 256, 768, 5, 448), (640, 2304, 768, 2, 320)])
def test_gemm_tn_big(Mtot, N, K, splits, rps):
    """256x256 pipeline TN kernel: 1, 2, 3, many t-steps; uneven / empty tails of the row split."""
    A, Bm = randbf(Mtot, N, seed=31), randbf(Mtot, K, seed=32)
    out = gemm_tn(A, Bm, N, splits, rps, big=True)
    ref = A.float().T @ Bm.float()
    assert rel_l2(out, ref) < 1e-5


@pytest.fixture
def bwd_form(request):
    """Forces the attention backward's form: 0 = dq + dkv kernels, 1 = the single-kernel form (S <= 512); afterwards the
    per-shape policy (-1) is back."""
    L = _lib.lib()
    L.plb_set_attn_bwd_fused(int(request.param))
    yield int(request.param)
    L.plb_set_attn_bwd_fused(-1)


@pytest.mark.parametrize("bwd_form", [1, 0], indirect=True)
@pytest.mark.parametrize("B,S,NH,lens", [(3, 40, 2, [40, 33, 7]), (2, 512, 3, [512, 300]), (2, 130, 2, None),
                                         (1, 64, 1, [1]), (3, 512, 2, [129, 256, 511]), (2, 257, 1, [257, 31]),
                                         (1, 600, 1, [600])])
def test_attention_fwd_bwd(B, S, NH, lens, bwd_form):
    L = _lib.lib()
    H = NH * 64
    qkv = randbf(B * S, 3 * H, scale=1.0, seed=9)
    lengths = torch.tensor(lens, dtype=torch.int32, device=DEV) if lens else None
    p, ctx, lse = attn_args(qkv, lengths, B, S, NH)
    assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
    torch.cuda.synchronize()
    rctx, rlse, grad = torch_attention(qkv, lengths, B, S, NH)
    assert rel_l2(ctx.float(), rctx) < 6e-3
    # the kernels keep MINUS the log-sum-exp in units of raw scores (the backward starts its accumulators there)
    assert (-lse * p.scale - rlse).abs().max() < 2e-3
    # backward: dctx zero on padded queries (as in the model), random elsewhere
    dctx = randbf(B * S, H, seed=10)
    if lens:
        qmask = (torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S, 1)
        dctx = dctx * qmask.to(dctx.dtype)
    delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
    dqkv = torch.full((B * S, 3 * H), 7.0, dtype=torch.bfloat16, device=DEV)
    p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
    assert L.plb_launch_attn_bwd(C.byref(p), stream()) == 0
    torch.cuda.synchronize()
    ref = grad(dctx)
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        if float(ref[:, sl].abs().max()) == 0.0:
            # a single valid key: P = 1 and dP - delta = 0 mathematically; the kernels form it as (-delta + sum of
            # products) in fp32, i.e. to rounding (1e-7 of the terms), not bit-exactly
            assert float(dqkv[:, sl].float().abs().max()) < 1e-5, name
        else:
            assert rel_l2(dqkv[:, sl].float(), ref[:, sl]) < 1.5e-2, name
    if lens:  # padded keys get exactly zero dK, dV — and padded queries exactly zero dQ
        kpad = ~(torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S)
        assert (dqkv[kpad] == 0).all()


@pytest.mark.parametrize("bwd_form", [1, 0], indirect=True)
@pytest.mark.parametrize("B,S,NH,lens", [(2, 512, 2, [512, 300]), (2, 130, 3, [130, 77]), (1, 64, 1, None)])
def test_attention_bwd_bias_gradient_partials(B, S, NH, lens, bwd_form):
    """colpart: the rows the backward leaves for the Q/K/V bias gradient sum to the column sums of the dqkv it stored,
    for the overwrite call (last layer) and the accumulating calls (the other applications of the shared layer)."""
    L = _lib.lib()
    H = NH * 64
    QT = (S + 127) // 128
    qkv = randbf(B * S, 3 * H, scale=1.0, seed=19)
    lengths = torch.tensor(lens, dtype=torch.int32, device=DEV) if lens else None
    p, ctx, lse = attn_args(qkv, lengths, B, S, NH)
    assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
    colp = torch.full((B * QT * 4, 3 * H), 3.0, dtype=torch.float32, device=DEV)
    delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
    total = torch.zeros(3 * H, dtype=torch.float64, device=DEV)
    for call in range(3):
        dctx = randbf(B * S, H, seed=40 + call)
        if lens:
            qmask = (torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S, 1)
            dctx = dctx * qmask.to(dctx.dtype)
        dqkv = torch.zeros((B * S, 3 * H), dtype=torch.bfloat16, device=DEV)
        p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
        p.colpart, p.colpart_accumulate = colp.data_ptr(), int(call > 0)
        assert L.plb_launch_attn_bwd(C.byref(p), stream()) == 0
        torch.cuda.synchronize()
        total += dqkv.double().sum(0)
        got = colp.double().sum(0)
        assert torch.allclose(got, total, rtol=1e-5, atol=1e-3 * float(total.abs().max())), call


@pytest.mark.parametrize("bwd_form", [0, 1], indirect=True)
def test_attention_race_screen(bwd_form):
    """The attention kernels hand LDS stages between LDS-DMA (global_load_lds, waited by vmcnt) and fragment reads across
    one barrier per tile: a mistake there shows as rare wrong tiles. Repeat launches must be bitwise identical."""
    L = _lib.lib()
    B, S, NH = 4, 512, 12
    H = NH * 64
    qkv = randbf(B * S, 3 * H, scale=1.0, seed=21)
    lengths = torch.tensor([512, 449, 130, 65], dtype=torch.int32, device=DEV)
    p, ctx, lse = attn_args(qkv, lengths, B, S, NH)
    dctx = randbf(B * S, H, seed=22)
    qmask = (torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S, 1)
    dctx = dctx * qmask.to(dctx.dtype)
    delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
    dqkv = torch.zeros((B * S, 3 * H), dtype=torch.bfloat16, device=DEV)
    p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
    first = None
    for _ in range(40):
        ctx.zero_(); dqkv.zero_()
        assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
        assert L.plb_launch_attn_bwd(C.byref(p), stream()) == 0
        torch.cuda.synchronize()
        got = (ctx.clone(), lse.clone(), dqkv.clone())
        if first is None:
            first = got
            rctx, _, grad = torch_attention(qkv, lengths, B, S, NH)
            valid = qmask.reshape(-1)
            assert rel_l2(ctx.float()[valid], rctx[valid]) < 6e-3
            assert rel_l2(dqkv.float(), grad(dctx)) < 1.5e-2
        else:
            assert all(torch.equal(a, b) for a, b in zip(got, first))


@pytest.mark.parametrize("B,NH,policy,nf", [(16, 16, -1, 16), (32, 12, -1, 0), (8, 12, -1, 0), (32, 12, 2, 21), (22, 12, 2, 21),
                                             (8, 12, 2, 0)])
def test_attention_bwd_policy_and_hybrid_split(B, NH, policy, nf):
    """The per-shape policy (attn.hip: plb_launch_attn_bwd); nf = samples the single-kernel form must take. Policy -1 (auto,
    the default): B x heads that fills the CUs in whole rounds takes it entirely (16 x 16 = 256 items = one round),
    everything else the two kernels. Policy 2 (auto + hybrid; measured slower at config A and off by default): a batch with
    at least one (nearly) full round of whole samples and a short remainder is SPLIT BY SAMPLE between the two forms (32 x
    12: 21 samples = 252 items fused, 11 two-kernel; 22 x 12: 21 + 1); small grids take the two kernels (8 x 12 = 96 items). Whatever the policy picks must equal the forced two-kernel result on the samples each form handled —
    the fused rows bitwise equal to a forced-fused call, the others bitwise equal to a forced-split call — including the
    bias-gradient partial rows."""
    L = _lib.lib()
    S = 512
    H = NH * 64
    QT = 4
    qkv = randbf(B * S, 3 * H, scale=1.0, seed=51)
    lengths = torch.full((B,), S, dtype=torch.int32, device=DEV)
    lengths[B - 1] = 300
    lengths[0] = 449
    p, ctx, lse = attn_args(qkv, lengths, B, S, NH)
    assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
    dctx = randbf(B * S, H, seed=52)
    qmask = (torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S, 1)
    dctx = dctx * qmask.to(dctx.dtype)
    delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
    outs = {}
    for form in (1, 0, policy):
        dqkv = torch.full((B * S, 3 * H), 7.0, dtype=torch.bfloat16, device=DEV)
        colp = torch.full((B * QT * 4, 3 * H), 3.0, dtype=torch.float32, device=DEV)
        p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
        p.colpart, p.colpart_accumulate = colp.data_ptr(), 0
        L.plb_set_attn_bwd_fused(form)
        try:
            _lib.profile_enable(True)
            assert L.plb_launch_attn_bwd(C.byref(p), stream()) == 0
            torch.cuda.synchronize()
            prof = _lib.profile_read()
        finally:
            _lib.profile_enable(False)
            L.plb_set_attn_bwd_fused(-1)
        outs[form] = (dqkv, colp, {k: v["launches"] for k, v in prof.items()})
    outs[-1] = outs[policy]
    launches = outs[-1][2]
    assert (launches.get("attn_bwd", 0), launches.get("attn_bwd_dq", 0)) == (int(nf > 0), int(nf < B)), (launches, nf)
    rows_f = slice(0, nf * S)
    rows_s = slice(nf * S, B * S)
    assert torch.equal(outs[-1][0][rows_f], outs[1][0][rows_f]) and torch.equal(outs[-1][0][rows_s], outs[0][0][rows_s])
    cf, cs = slice(0, nf * QT * 4), slice(nf * QT * 4, B * QT * 4)
    assert torch.equal(outs[-1][1][cf], outs[1][1][cf]) and torch.equal(outs[-1][1][cs], outs[0][1][cs])
    ref = torch_attention(qkv, lengths, B, S, NH)[2](dctx)
    assert rel_l2(outs[-1][0].float(), ref) < 1.5e-2
    assert torch.allclose(outs[-1][1].double().sum(0), outs[-1][0].double().sum(0), rtol=1e-5, atol=1e-3 * float(ref.abs().max()) * 10)


@pytest.mark.parametrize("with_rows", [False, True])
def test_attention_bwd_fused_writes_the_fp8_image(with_rows):
    """fp8 calls: the single-kernel form writes the e5m2 image of dQKV itself (alone in a training call whose weight
    gradients read images: p.dqkv = NULL) — the values as rounded to bf16, times the site's scale — and reports the
    maximum. A call that needs rows AND image (with_rows) is refused by the fused launcher and taken by the two kernels;
    the images of the two forms agree wherever their bf16 rows agree bitwise, and within an e5m2 step elsewhere."""
    L = _lib.lib()
    B, S, NH = 4, 512, 4
    H = NH * 64
    qkv = randbf(B * S, 3 * H, scale=1.0, seed=61)
    lengths = torch.tensor([512, 512, 400, 77], dtype=torch.int32, device=DEV)
    p, ctx, lse = attn_args(qkv, lengths, B, S, NH)
    assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
    dctx = randbf(B * S, H, seed=62)
    qmask = (torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S, 1)
    dctx = dctx * qmask.to(dctx.dtype)
    delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
    scale = torch.tensor([512.0], device=DEV)
    res = {}
    for form in (1, 0):
        dqkv = torch.zeros((B * S, 3 * H), dtype=torch.bfloat16, device=DEV)
        img = torch.full((B * S, 3 * H), 0x7B, dtype=torch.uint8, device=DEV)
        amax = torch.zeros(64 * 16, device=DEV)
        p.dctx, p.lddctx, p.delta, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), 3 * H
        p.dqkv = dqkv.data_ptr() if (with_rows or form == 0) else None
        p.dqkv8, p.lddqkv8, p.dqkv_scale, p.dqkv_amax = img.data_ptr(), 3 * H, scale.data_ptr(), amax.data_ptr()
        if form == 1 and with_rows:
            assert L.plb_launch_attn_bwd_fused(C.byref(p), stream()) != 0     # refused: the policy sends it to the two kernels
            continue
        L.plb_set_attn_bwd_fused(form)
        try:
            assert L.plb_launch_attn_bwd(C.byref(p), stream()) == 0
            torch.cuda.synchronize()
        finally:
            L.plb_set_attn_bwd_fused(-1)
        res[form] = (dqkv, img.view(torch.float8_e5m2).float() / 512.0, float(amax.max()))
    if with_rows:
        return
    rows, img0, am0 = res[0]
    _, img1, am1 = res[1]
    want = (rows.float() * 512.0).clamp(-57344, 57344).to(torch.float8_e5m2).float() / 512.0
    assert torch.equal(img0, want)                                  # two-kernel form: image of its own bf16 rows
    assert abs(am0 - float(rows.float().abs().max())) <= 1e-6 * am0
    # the fused form's values differ from the two-kernel form's by accumulation order (bf16 rounding flips), so its image is
    # compared as numbers: within the e5m2 step of the reference gradient, and its maximum is the maximum of what it stored
    ref = torch_attention(qkv, lengths, B, S, NH)[2](dctx)
    assert rel_l2(img1, ref) < 0.08 and rel_l2(img0, ref) < 0.08   # e5m2: 2 mantissa bits
    assert abs(am1 - am0) <= 2e-2 * am0
    kpad = ~(torch.arange(S, device=DEV)[None, :] < lengths[:, None]).reshape(B * S)
    assert (img1[kpad] == 0).all()


@pytest.mark.parametrize("B,S,NH,lens,counts", [(3, 512, 2, [512, 300, 512], [70, 33, 140]), (2, 130, 3, None, [5, 0]),
                                                (4, 512, 12, [512, 449, 130, 512], [64, 128, 1, 200])])
def test_attention_compact_queries(B, S, NH, lens, counts):
    """Compact-query mode (PlbAttn.qoff): the queries of a sample are a LIST of rows (the masked positions of the last
    application, csrc/engine.cpp) gathered into a compact buffer, keys and values stay all S rows. Forward: the compact
    context rows and statistics equal the full evaluation's rows at those positions BITWISE (a query's arithmetic does not
    depend on which other queries share its tile). Backward with dO given on the compact rows only: dQ of the compact rows,
    dK / dV of all rows and the bias-gradient partial rows against torch autograd of the full attention with dO zero
    elsewhere — and against the full two-kernel form on the scattered dO. Counts include an empty sample, a single query,
    exactly one tile and more than one tile."""
    L = _lib.lib()
    H = NH * 64
    QT = (S + 127) // 128
    qkv = randbf(B * S, 3 * H, scale=1.0, seed=71)
    lengths = torch.tensor(lens, dtype=torch.int32, device=DEV) if lens else None
    g = torch.Generator().manual_seed(5)
    rows, off = [], [0]
    for b, n in enumerate(counts):
        lim = lens[b] if lens else S
        pos = torch.randperm(lim, generator=g)[:n].sort().values
        rows += (pos + b * S).tolist()
        off.append(off[-1] + n)
    Nq = off[-1]
    rows_t = torch.tensor(rows, dtype=torch.int64, device=DEV)
    qoff = torch.tensor(off, dtype=torch.int32, device=DEV)
    qc = qkv[rows_t, :H].contiguous()
    # the full evaluation
    p, ctx, lse = attn_args(qkv, lengths, B, S, NH)
    assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
    # compact queries
    pc, _, _ = attn_args(qkv, lengths, B, S, NH)
    ctxc = torch.zeros((max(Nq, 1), H), dtype=torch.bfloat16, device=DEV)
    lsec = torch.zeros((NH, max(Nq, 1)), dtype=torch.float32, device=DEV)
    pc.ctx, pc.ldctx, pc.lse = ctxc.data_ptr(), H, lsec.data_ptr()
    pc.qoff, pc.q, pc.ldq, pc.nq_total = qoff.data_ptr(), qc.data_ptr(), H, Nq
    assert L.plb_launch_attn_fwd(C.byref(pc), stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(ctxc[:Nq], ctx[rows_t])
    full_lse = lse.reshape(B, NH, S).permute(1, 0, 2).reshape(NH, B * S)[:, rows_t]
    assert torch.equal(lsec[:, :Nq], full_lse)
    # backward: dO on the compact rows
    dctxc = randbf(max(Nq, 1), H, seed=72)
    dctx_full = torch.zeros((B * S, H), dtype=torch.bfloat16, device=DEV)
    dctx_full[rows_t] = dctxc[:Nq]
    out = {}
    for mode in ("compact", "full"):
        dqkv = torch.full((B * S, 3 * H), 7.0, dtype=torch.bfloat16, device=DEV)
        colp = torch.full((B * QT * 4, 3 * H), 3.0, dtype=torch.float32, device=DEV)
        q = pc if mode == "compact" else p
        if mode == "compact":
            delta = torch.zeros((NH, max(Nq, 1)), dtype=torch.float32, device=DEV)
            dqc = torch.full((max(Nq, 1), H), 5.0, dtype=torch.bfloat16, device=DEV)
            q.dctx, q.lddctx, q.dq, q.lddq = dctxc.data_ptr(), H, dqc.data_ptr(), H
        else:
            delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
            q.dctx, q.lddctx = dctx_full.data_ptr(), H
        q.delta, q.dqkv, q.lddqkv, q.colpart, q.colpart_accumulate = delta.data_ptr(), dqkv.data_ptr(), 3 * H, colp.data_ptr(), 0
        L.plb_set_attn_bwd_fused(0)
        try:
            assert L.plb_launch_attn_bwd(C.byref(q), stream()) == 0
            torch.cuda.synchronize()
        finally:
            L.plb_set_attn_bwd_fused(-1)
        out[mode] = (dqkv, colp, dqc if mode == "compact" else None)
    ref = torch_attention(qkv, lengths, B, S, NH)[2](dctx_full)
    dq_c, (dqkv_c, colp_c, dqc) = None, out["compact"]
    dqkv_f, colp_f, _ = out["full"]
    # dK, dV of every row: the same products, but the MFMA sums group the non-zero terms differently when the zero rows
    # between them are gone: equal to fp32 summation noise seen through bf16 rounding, not bitwise
    if Nq:
        assert rel_l2(dqkv_c[:, H:].float(), dqkv_f[:, H:].float()) < 4e-3
    assert (dqkv_c[:, :H] == 7.0).all()                                  # the Q block of dqkv is not touched in compact mode
    assert torch.equal(dqc[:Nq], dqkv_f[rows_t, :H])                     # dQ of the compact rows = the full form's rows
    if Nq:
        assert rel_l2(dqc[:Nq].float(), ref[rows_t, :H]) < 1.5e-2
        assert rel_l2(dqkv_c[:, H:].float(), ref[:, H:]) < 1.5e-2
    # bias-gradient partial rows: K / V columns as in the full form, Q columns sum to the column sums of the compact dQ
    assert torch.allclose(colp_c[:, H:].double().sum(0), dqkv_c[:, H:].double().sum(0), rtol=1e-5, atol=1e-3 * (float(ref.abs().max()) + 1e-6) * 10)
    assert torch.allclose(colp_c[:, :H].double().sum(0), dqc[:Nq].double().sum(0), rtol=1e-5, atol=1e-3 * (float(ref.abs().max()) + 1e-6) * 10)


@pytest.mark.parametrize("ramp", [0.0, 0.02, 0.5])
def test_attention_running_maximum(ramp):
    """The forward rescales its accumulators only when a row maximum has grown by more than 2^8 and keeps a stale
    reference otherwise. Keys whose scores grow along the sequence exercise both: ramp 0.5 moves the maximum by tens of
    log2 units per 64-key tile (a rescale in every tile), 0.02 by a fraction of the threshold (stale reference with
    probabilities above 1), 0 is the flat case. Forward, LSE and backward against fp32 torch."""
    L = _lib.lib()
    B, S, NH = 2, 512, 2
    H = NH * 64
    g = torch.Generator(device="cpu").manual_seed(77)
    q = torch.randn(B * S, H, generator=g)
    k = torch.randn(B * S, H, generator=g) * 0.3
    v = torch.randn(B * S, H, generator=g)
    # key j gets a component along the query's mean direction that grows with j: scores rise along the keys
    qdir = torch.nn.functional.normalize(q.reshape(B, S, NH, 64).mean(1, keepdim=True), dim=-1)      # [B,1,NH,64]
    pos = torch.arange(S, dtype=torch.float32).reshape(1, S, 1, 1)
    k = (k.reshape(B, S, NH, 64) + ramp * pos * qdir).reshape(B * S, H)
    q = (q.reshape(B, S, NH, 64) + 3.0 * qdir).reshape(B * S, H)                                       # queries lean that way
    qkv = torch.cat([q, k, v], dim=1).to(torch.bfloat16).to(DEV).contiguous()
    p, ctx, lse = attn_args(qkv, None, B, S, NH)
    assert L.plb_launch_attn_fwd(C.byref(p), stream()) == 0
    torch.cuda.synchronize()
    rctx, rlse, grad = torch_attention(qkv, None, B, S, NH)
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(lse).all()
    assert rel_l2(ctx.float(), rctx) < 6e-3
    assert (-lse * p.scale - rlse).abs().max() < 2e-3 * max(1.0, float(rlse.abs().max()) / 10)
    dctx = randbf(B * S, H, seed=11)
    delta = torch.zeros((B, NH, S), dtype=torch.float32, device=DEV)
    dqkv = torch.zeros((B * S, 3 * H), dtype=torch.bfloat16, device=DEV)
    p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
    assert L.plb_launch_attn_bwd(C.byref(p), stream()) == 0
    torch.cuda.synchronize()
    ref = grad(dctx)
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        # at ramp 0.5 the keys reach |k| = 250 along one direction and sum_k dS = 0 only in exact arithmetic: dq / dk are
        # then differences of bf16-rounded terms 250x their size (ill-conditioned for any bf16 kernel); dv is not
        if ramp < 0.1 or name == "dv":
            assert rel_l2(dqkv[:, sl].float(), ref[:, sl]) < 2e-2, name


@pytest.mark.parametrize("T,H", [(37, 128), (300, 768), (64, 1024)])
def test_layernorm_fwd_bwd(T, H):
    L = _lib.lib()
    x = randbf(T + 3, H, scale=2.0, seed=11)
    gam = (1 + 0.1 * torch.randn(H, device=DEV)).contiguous()
    bet = (0.1 * torch.randn(H, device=DEV)).contiguous()
    y = torch.zeros((T + 3, H), dtype=torch.bfloat16, device=DEV)
    mean = torch.zeros(T + 3, device=DEV)
    rstd = torch.zeros(T + 3, device=DEV)
    p = _lib.PlbLayerNorm()
    p.x, p.ldx, p.gamma, p.beta, p.eps = x.data_ptr(), H, gam.data_ptr(), bet.data_ptr(), 1e-12
    p.y, p.ldy, p.mean, p.rstd, p.T, p.H, p.Tzero = y.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), T, H, T + 3
    assert L.plb_launch_ln_fwd(C.byref(p), stream()) == 0
    xr = x[:T].float().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (H,), gam, bet, eps=1e-12)
    torch.cuda.synchronize()
    assert rel_l2(y[:T].float(), yr.detach()) < 4e-3
    dy = randbf(T + 3, H, seed=12)
    dx = torch.full((T + 3, H), 3.0, dtype=torch.bfloat16, device=DEV)
    nb = 64
    part = torch.zeros((nb, 3 * H), device=DEV)
    p.dy, p.lddy, p.dx, p.lddx, p.partials, p.nblocks = dy.data_ptr(), H, dx.data_ptr(), H, part.data_ptr(), nb
    assert L.plb_launch_ln_bwd(C.byref(p), stream()) == 0
    torch.cuda.synchronize()
    gw = torch.nn.Parameter(gam.clone())
    gb = torch.nn.Parameter(bet.clone())
    xr2 = x[:T].float().clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr2, (H,), gw, gb, eps=1e-12).backward(dy[:T].float())
    assert rel_l2(dx[:T].float(), xr2.grad) < 5e-3
    assert (dx[T:] == 0).all()
    assert rel_l2(part.sum(0)[:H], gw.grad) < 1e-4
    assert rel_l2(part.sum(0)[H:2 * H], gb.grad) < 1e-4
    # third block: column sums of dx as stored (the bias gradient of the Linear in front of the LayerNorm)
    assert rel_l2(part.sum(0)[2 * H:], dx[:T].float().sum(0)) < 1e-5


def test_colsum():
    L = _lib.lib()
    R, N = 1000, 776
    x = randbf(R, 784, seed=13)
    out = torch.zeros(N - 4, device=DEV)
    scratch = torch.zeros(16 * N, device=DEV)
    assert L.plb_launch_colsum(x.data_ptr(), 1, R, N, 784, out.data_ptr(), N - 4, 0, scratch.data_ptr(), 16, stream()) == 0
    torch.cuda.synchronize()
    assert rel_l2(out, x[:, : N - 4].float().sum(0)) < 1e-5


@pytest.mark.parametrize("NT,B,S,lens,tile", [(1000, 2, 64, [64, 7], 1256), (256, 2, 128, None, 256), (4100, 1, 130, [77], 1256)])
def test_fused_gemm_cross_entropy_passes(NT, B, S, lens, tile):
    """The two GEMM passes of the fused token-head CE + the merge kernel against torch: loss rows with weight
    1 / (B * len), gradient (softmax - onehot) * w in bf16, zeros on padded positions, on rows >= B*S and on columns
    >= NT, and the column-sum partials (bias gradient) of the gradient as stored."""
    L = _lib.lib()
    L.plb_launch_gemm_nt_big.restype = C.c_int
    L.plb_launch_gemm_nt_big.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.plb_launch_token_ce_combine.restype = C.c_int
    L.plb_launch_token_ce_combine.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    T, H = B * S, 128
    Tp = (T + 255) // 256 * 256 if tile == 256 else (T + 127) // 128 * 128
    NTp = (NT + 255) // 256 * 256
    g = torch.Generator(device=DEV).manual_seed(11)
    A = (torch.randn(Tp, H, device=DEV, generator=g)).to(torch.bfloat16)
    W = (torch.randn(NTp, H, device=DEV, generator=g) * 0.3).to(torch.bfloat16)
    bias = torch.zeros(NTp, device=DEV)
    bias[:NT] = torch.randn(NT, device=DEV, generator=g)
    tgt = torch.zeros(Tp, dtype=torch.int64, device=DEV)
    tgt[:T] = torch.randint(0, NT, (T,), device=DEV, generator=g)
    ntile = NTp // 256
    pmax, psum = torch.empty(Tp, ntile, device=DEV), torch.empty(Tp, ntile, device=DEV)
    tl, lse, w, lrows = (torch.full((Tp,), 7.0, device=DEV) for _ in range(4))
    dl = torch.full((Tp, NTp), 3.0, dtype=torch.bfloat16, device=DEV)
    cprows = 2 * (Tp // 256) if tile == 256 else 2 * (Tp // 128)
    colp = torch.empty(cprows, NTp, device=DEV)
    lengths = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=DEV)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb, p.M, p.N, p.K, p.Mstore = A.data_ptr(), H, W.data_ptr(), H, Tp, NTp, H, Tp
    p.bias, p.ce_cols, p.ce_tgt = bias.data_ptr(), NT, tgt.data_ptr()
    p.ce_pmax, p.ce_psum, p.ce_tlogit = pmax.data_ptr(), psum.data_ptr(), tl.data_ptr()
    assert L.plb_launch_gemm_nt_big(C.byref(p), tile, 3, 0, stream()) == 0
    assert L.plb_launch_token_ce_combine(pmax.data_ptr(), psum.data_ptr(), ntile, tl.data_ptr(),
                                         None if lengths is None else lengths.data_ptr(), B, S, Tp, lse.data_ptr(),
                                         w.data_ptr(), lrows.data_ptr(), stream()) == 0
    p.ce_lse, p.ce_w, p.C, p.ldc, p.colpart = lse.data_ptr(), w.data_ptr(), dl.data_ptr(), NTp, colp.data_ptr()
    assert L.plb_launch_gemm_nt_big(C.byref(p), tile, 4, 0, stream()) == 0
    torch.cuda.synchronize()
    ln = torch.tensor(lens if lens is not None else [S] * B, device=DEV)
    valid = (torch.arange(S, device=DEV)[None, :] < ln[:, None]).reshape(-1)
    wr = (1.0 / (B * ln.double()))[:, None].expand(B, S).reshape(-1)
    x = (A[:T].double() @ W[:NT].double().T) + bias[:NT].double()
    ref_lse = torch.logsumexp(x, -1)
    want_loss = torch.where(valid, wr * (ref_lse - x[torch.arange(T), tgt[:T]]), torch.zeros_like(ref_lse))
    assert torch.allclose(lrows[:T].double(), want_loss, rtol=2e-5, atol=1e-7)
    assert (lrows[T:] == 0).all() and (w[T:] == 0).all()
    pr = torch.softmax(x, -1)
    pr[torch.arange(T), tgt[:T]] -= 1.0
    want = pr * wr[:, None] * valid[:, None]
    assert rel_l2(dl[:T, :NT].float(), want.float()) < 4e-3                 # one bf16 rounding
    assert (dl[:T, NT:] == 0).all() and (dl[T:] == 0).all() and (dl[:T][~valid] == 0).all()
    assert rel_l2(colp.sum(0)[:NT], dl[:, :NT].float().sum(0)) < 1e-5        # bias gradient = column sums as stored
