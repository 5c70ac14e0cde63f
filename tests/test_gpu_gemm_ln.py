"""GEMM with LayerNorm in the epilogue (-m gpu; csrc/gemm_ln.hip, gemm_nt_pipeline.h ACT 5 / 6): the dense / FFN-output
projections + AlbertAttention.LayerNorm / full_layer_layer_norm forward (modeling_albert.py:196-200, 225-238) and the
backward of those LayerNorms inside the dX GEMM that produces their output gradient. The column tiles of a row block
exchange row partials INSIDE the launch (agent-scope hand-off): besides the arithmetic (against fp32 torch on the same
bf16 operands, and bit for bit against the unfused GEMM for the stored pre-LayerNorm sums) the tests check the hand-off:
flags back to zero, no time-out, repeated launches bitwise identical with the consumers' caches warm."""
import ctypes as C

import pytest
import torch

from gpu_util import gemm_nt, rel_l2, stream
from plbert_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


def randbf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


class Ln:
    """Buffers of one fused launch."""

    def __init__(self, M, N, K, seed=0):
        self.M, self.N, self.K = M, N, K
        self.A, self.B = randbf(M, K, seed=seed + 1), randbf(N, K, scale=K ** -0.5, seed=seed + 2)
        self.bias = torch.randn(N, generator=torch.Generator().manual_seed(seed + 3)).to(DEV)
        self.res = randbf(M, N, seed=seed + 4)
        g = torch.Generator().manual_seed(seed + 5)
        self.gamma = (1.0 + 0.3 * torch.randn(N, generator=g)).to(DEV)
        self.beta = (0.2 * torch.randn(N, generator=g)).to(DEV)
        nbn = N // (384 if N % 384 == 0 else 256)
        self.nbn = nbn
        self.xchg = torch.zeros(M // 128 * nbn * nbn * 128 * 2, dtype=torch.int64, device=DEV)
        self.err = torch.zeros(1, dtype=torch.int32, device=DEV)
        self.mean = torch.zeros(M, dtype=torch.float32, device=DEV)
        self.rstd = torch.zeros(M, dtype=torch.float32, device=DEV)

    def params(self):
        p = _lib.PlbGemmNT()
        p.A, p.lda, p.B, p.ldb = self.A.data_ptr(), self.K, self.B.data_ptr(), self.K
        p.M, p.N, p.K, p.Mstore = self.M, self.N, self.K, self.M
        p.ln_gamma, p.ln_beta, p.ln_mean, p.ln_rstd = self.gamma.data_ptr(), self.beta.data_ptr(), self.mean.data_ptr(), self.rstd.data_ptr()
        p.ln_eps = 1e-12
        p.ln_xchg, p.ln_err = self.xchg.data_ptr(), self.err.data_ptr()
        return p


def _fwd(t, reps=1):
    L = _lib.lib()
    pre = torch.zeros(t.M, t.N, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros_like(pre)
    p = t.params()
    p.bias, p.res, p.ldr = t.bias.data_ptr(), t.res.data_ptr(), t.N
    p.C, p.ldc, p.C2, p.ldc2 = pre.data_ptr(), t.N, y.data_ptr(), t.N
    outs = []
    for _ in range(reps):
        assert L.plb_launch_gemm_nt_ln(C.byref(p), 5, stream()) == 0
        torch.cuda.synchronize()
        outs.append((pre.clone(), y.clone(), t.mean.clone(), t.rstd.clone()))
    return outs


@pytest.mark.parametrize("M,N,K", [(1024, 768, 768), (2048, 768, 2048), (1024, 1024, 1024), (16384, 768, 768),
                                   (1024, 768, 64), (1024, 768, 128), (1024, 768, 192), (1024, 1024, 64), (1024, 1024, 128)])
def test_gemm_layernorm_forward(M, N, K):
    t = Ln(M, N, K, seed=M + N)
    pre, y, mean, rstd = _fwd(t)[0]
    ref_pre, _ = gemm_nt(t.A, t.B, N, bias=t.bias, res=t.res)
    assert torch.equal(pre, ref_pre)                                        # the stored sum is the unfused GEMM's, bit for bit
    x = pre.float()
    mu, var = x.mean(1), x.var(1, unbiased=False)
    assert (mean - mu).abs().max() < 1e-5 * max(1.0, float(mu.abs().max()))
    assert ((rstd - (var + 1e-12).rsqrt()) / rstd).abs().max() < 1e-5
    ref_y = torch.nn.functional.layer_norm(x, (N,), t.gamma, t.beta, 1e-12)
    assert rel_l2(y.float(), ref_y) < 3e-3
    assert int(t.err.item()) == 0 and int(t.xchg.abs().sum().item()) == 0  # no time-out; every hand-off word cleared


def _bwd(t, reps=1):
    """dy = bf16(A·B^T + res) is the gradient of the LayerNorm output; aux = pre of a forward over other operands."""
    L = _lib.lib()
    pre = randbf(t.M, t.N, seed=77)
    x = pre.float()
    t.mean.copy_(x.mean(1)); t.rstd.copy_((x.var(1, unbiased=False) + 1e-12).rsqrt())
    dx = torch.zeros(t.M, t.N, dtype=torch.bfloat16, device=DEV)
    colp = torch.full((2 * t.M // 128, 3, t.N), 9.0, dtype=torch.float32, device=DEV)
    p = t.params()
    p.res, p.ldr, p.aux, p.ldaux = t.res.data_ptr(), t.N, pre.data_ptr(), t.N
    p.C, p.ldc, p.colpart = dx.data_ptr(), t.N, colp.data_ptr()
    outs = []
    for _ in range(reps):
        assert L.plb_launch_gemm_nt_ln(C.byref(p), 6, stream()) == 0
        torch.cuda.synchronize()
        outs.append((dx.clone(), colp.clone()))
    return pre, outs


@pytest.mark.parametrize("M,N,K", [(1024, 768, 2048), (2048, 768, 2304), (1024, 1024, 4096), (16384, 768, 2048)])
def test_gemm_layernorm_backward(M, N, K):
    t = Ln(M, N, K, seed=M + K)
    pre, outs = _bwd(t)
    dx, colp = outs[0]
    dy, _ = gemm_nt(t.A, t.B, N, res=t.res)                                  # what the unfused path stores and reads back
    x = pre.float().requires_grad_(True)
    gamma = t.gamma.clone().requires_grad_(True)
    beta = t.beta.clone().requires_grad_(True)
    yy = torch.nn.functional.layer_norm(x, (N,), gamma, beta, 1e-12)
    yy.backward(dy.float())
    assert rel_l2(dx.float(), x.grad) < 4e-3
    sums = colp.double().sum(0)
    assert rel_l2(sums[0], gamma.grad.double()) < 1e-4                       # dgamma
    assert rel_l2(sums[1], beta.grad.double()) < 1e-4                        # dbeta
    assert rel_l2(sums[2], dx.double().sum(0)) < 1e-5                        # column sums of dx as stored (bias gradient)
    assert int(t.err.item()) == 0 and int(t.xchg.abs().sum().item()) == 0


@pytest.mark.parametrize("mode", [5, 6])
def test_rows_past_mstore_are_computed_but_never_stored(mode):
    """Mstore < M (the engine's padded token rows): every output row below Mstore equals the full launch's, every row at or
    above it keeps its sentinel — images, statistics and the lane-private operand staging must not leak past the bound."""
    L = _lib.lib()
    M, N, K, ms = 2048, 768, 768, 2048 - 300
    t = Ln(M, N, K, seed=11)
    full = _fwd(t)[0] if mode == 5 else _bwd(t)[1][0]
    p = t.params()
    p.Mstore = ms
    if mode == 5:
        pre = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        y = torch.full_like(pre, 7.0)
        t.mean.fill_(7.0); t.rstd.fill_(7.0)
        p.bias, p.res, p.ldr = t.bias.data_ptr(), t.res.data_ptr(), N
        p.C, p.ldc, p.C2, p.ldc2 = pre.data_ptr(), N, y.data_ptr(), N
        assert L.plb_launch_gemm_nt_ln(C.byref(p), 5, stream()) == 0
        torch.cuda.synchronize()
        outs = (pre, y, t.mean.clone(), t.rstd.clone())
        for o, f in zip(outs, full):
            assert torch.equal(o[:ms], f[:ms]) and bool((o[ms:] == 7.0).all())
    else:
        aux = randbf(M, N, seed=77)
        x = aux.float()
        t.mean.copy_(x.mean(1)); t.rstd.copy_((x.var(1, unbiased=False) + 1e-12).rsqrt())
        dx = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        colp = torch.zeros(2 * M // 128, 3, N, dtype=torch.float32, device=DEV)
        p.res, p.ldr, p.aux, p.ldaux = t.res.data_ptr(), N, aux.data_ptr(), N
        p.C, p.ldc, p.colpart = dx.data_ptr(), N, colp.data_ptr()
        assert L.plb_launch_gemm_nt_ln(C.byref(p), 6, stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(dx[:ms], full[0][:ms]) and bool((dx[ms:] == 7.0).all())
        # whole 128-row tiles below the bound leave the same partial rows; the column sums of dx count stored rows only
        nt = ms // 128
        assert torch.equal(colp[: 2 * nt], full[1][: 2 * nt])
        assert rel_l2(colp[:, 2].double().sum(0), dx[:ms].double().sum(0)) < 1e-5
    assert int(t.err.item()) == 0 and int(t.xchg.abs().sum().item()) == 0


@pytest.mark.parametrize("mode", [5, 6])
def test_hand_off_race_screen(mode):
    """The partner tiles' partials cross CUs (possibly XCDs) inside the launch: 60 launches over the SAME buffers — the
    consumers' caches hold the previous launch's lines — must agree bit for bit, at a shape with 2 and one with 4 column
    tiles per row block."""
    for (M, N, K) in ((4096, 768, 768), (2048, 1024, 1024)):
        t = Ln(M, N, K, seed=5)
        outs = _fwd(t, reps=60) if mode == 5 else _bwd(t, reps=60)[1]
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
        assert int(t.err.item()) == 0 and int(t.xchg.abs().sum().item()) == 0


def test_unsupported_shapes_are_refused():
    L = _lib.lib()
    t = Ln(1024, 768, 768)
    p = t.params()
    p.M = 896                                                                # not a multiple of 1024: no fused form
    assert L.plb_launch_gemm_nt_ln(C.byref(p), 5, stream()) == 3
    p.M = 1024                                                               # operand rows off 16-byte boundaries: refused
    pre, y = torch.zeros(1024, 768, dtype=torch.bfloat16, device=DEV), torch.zeros(1024, 768, dtype=torch.bfloat16, device=DEV)
    p.bias, p.C, p.ldc, p.C2, p.ldc2 = t.bias.data_ptr(), pre.data_ptr(), 768, y.data_ptr(), 768
    p.res, p.ldr = t.res.data_ptr() + 2, 768
    assert L.plb_launch_gemm_nt_ln(C.byref(p), 5, stream()) == 1
    p.res, p.ldr = t.res.data_ptr(), 772
    assert L.plb_launch_gemm_nt_ln(C.byref(p), 5, stream()) == 1


def lane_stash_to_rows(blob, M, N):
    """The gelu'(u) stash of forms 7 / 8 is in LANE layout (csrc/gemm_nt_pipeline.h): per 256x256 tile t (row-major tile
    order), slab (mh, mi) and column half nh, 512 consecutive 16-byte items, one per thread = {ni 0, ni 1} x 4 columns."""
    nbn = N // 256
    t = torch.arange((M // 256) * nbn, device=blob.device).view(-1, 1, 1, 1, 1, 1, 1)
    mh = torch.arange(2, device=blob.device).view(1, -1, 1, 1, 1, 1, 1)
    mi = torch.arange(4, device=blob.device).view(1, 1, -1, 1, 1, 1, 1)
    nh = torch.arange(2, device=blob.device).view(1, 1, 1, -1, 1, 1, 1)
    tid = torch.arange(512, device=blob.device).view(1, 1, 1, 1, -1, 1, 1)
    ni = torch.arange(2, device=blob.device).view(1, 1, 1, 1, 1, -1, 1)
    r = torch.arange(4, device=blob.device).view(1, 1, 1, 1, 1, 1, -1)
    uw, lane = tid >> 6, tid & 63
    wm, wn, frow, fq = uw >> 2, uw & 3, lane & 15, lane >> 4
    row = (t // nbn) * 256 + mh * 128 + wm * 64 + mi * 16 + frow
    col = (t % nbn) * 256 + nh * 128 + wn * 32 + ni * 16 + fq * 4 + r
    out = torch.empty(M, N, dtype=blob.dtype, device=blob.device)
    row, col = torch.broadcast_tensors(row, col)
    out[row.reshape(-1), col.reshape(-1)] = blob.reshape(-1)   # the blob's element order IS (t, mh, mi, nh, tid, ni, r)
    return out


def test_gelu_epilogues_with_stashed_derivative():
    """Forms 7 / 8 of the pipeline kernel (plb_launch_gemm_nt_gelud): the FFN up-projection writes gelu_new'(u) and
    gelu_new(u) (activations.py:59-66) from the fp32 u = A·B^T + bias; the backward GEMM multiplies by the stash and
    leaves the column-sum partials of its output (the ffn.bias gradient)."""
    import math
    L = _lib.lib()
    M, N, K = 768, 768, 256
    A, B = randbf(M, K, seed=4), randbf(N, K, scale=0.15, seed=5)
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(6)) * 0.1).to(DEV)
    d = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    g = torch.zeros_like(d)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb, p.M, p.N, p.K, p.Mstore = A.data_ptr(), K, B.data_ptr(), K, M, N, K, M
    p.bias, p.C, p.ldc, p.C2, p.ldc2 = bias.data_ptr(), d.data_ptr(), N, g.data_ptr(), N
    assert L.plb_launch_gemm_nt_gelud(C.byref(p), 0, stream()) == 0
    torch.cuda.synchronize()
    u = (A.float() @ B.float().T + bias).requires_grad_(True)
    act = 0.5 * u * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (u + 0.044715 * u ** 3)))
    act.sum().backward()
    assert rel_l2(g.float(), act.detach()) < 4e-3
    drows = lane_stash_to_rows(d, M, N)
    assert rel_l2(drows.float(), u.grad) < 4e-3 and (drows.float() - u.grad).abs().max() < 1e-2
    # backward: (A2·B2^T) * stash, + column sums of what was stored
    A2, B2 = randbf(M, K, seed=7), randbf(N, K, scale=0.15, seed=8)
    du = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    colp = torch.zeros(2 * M // 256, N, dtype=torch.float32, device=DEV)
    q = _lib.PlbGemmNT()
    q.A, q.lda, q.B, q.ldb, q.M, q.N, q.K, q.Mstore = A2.data_ptr(), K, B2.data_ptr(), K, M, N, K, M
    q.aux, q.ldaux, q.C, q.ldc, q.colpart = d.data_ptr(), N, du.data_ptr(), N, colp.data_ptr()
    assert L.plb_launch_gemm_nt_gelud(C.byref(q), 1, stream()) == 0
    torch.cuda.synchronize()
    ref = (A2.float() @ B2.float().T) * drows.float()
    assert rel_l2(du.float(), ref) < 4e-3
    assert rel_l2(colp.double().sum(0), du.double().sum(0)) < 1e-5
    q.M = 384                                                                # not a 256-multiple: no such form
    assert L.plb_launch_gemm_nt_gelud(C.byref(q), 1, stream()) == 3
