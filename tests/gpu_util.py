"""Helpers shared by the -m gpu tests: raw launches through the C ABI with torch tensors as buffers."""
import ctypes as C

import torch

from plbert_amd import _lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def gemm_nt(A, B, N, bias=None, res=None, aux=None, act=0, out_f32=False, Mstore=None, ldout=None):
    """A [M,K] bf16, B [>=ceil128(N),K] bf16 -> C (bf16 or fp32) [M, ldout]."""
    L = _lib.lib()
    M, K = A.shape
    ldout = ldout or N
    dev = A.device
    Cb = torch.zeros((M, ldout), dtype=torch.bfloat16, device=dev)
    C2 = torch.zeros((M, ldout), dtype=torch.bfloat16, device=dev)
    Cf = torch.zeros((M, ldout), dtype=torch.float32, device=dev)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0)
    p.M, p.N, p.K, p.Mstore = M, N, K, (M if Mstore is None else Mstore)
    p.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        p.res, p.ldr = res.data_ptr(), res.stride(0)
    if aux is not None:
        p.aux, p.ldaux = aux.data_ptr(), aux.stride(0)
    p.C, p.ldc, p.C2, p.ldc2, p.Cf, p.ldcf = Cb.data_ptr(), ldout, C2.data_ptr(), ldout, Cf.data_ptr(), ldout
    rc = L.plb_launch_gemm_nt(C.byref(p), act, int(out_f32), stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return (Cf if out_f32 else Cb), C2


def gemm_tn(A, B, N, splits, rows_per_split, big=False):
    """A [Mtot, Ncols] bf16, B [Mtot, K] bf16 -> dW [N,K] fp32 (slabs reduced)."""
    L = _lib.lib()
    Mtot, Ncols = A.shape
    K = B.shape[1]
    slab = torch.zeros((splits, N, K), dtype=torch.float32, device=A.device)
    out = torch.zeros((N, K), dtype=torch.float32, device=A.device)
    p = _lib.PlbGemmTN()
    p.A, p.lda, p.Ncols, p.B, p.ldb = A.data_ptr(), A.stride(0), Ncols, B.data_ptr(), B.stride(0)
    p.Mtot, p.N, p.K, p.rows_per_split, p.splits, p.slab = Mtot, N, K, rows_per_split, splits, slab.data_ptr()
    rc = (L.plb_launch_gemm_tn_big if big else L.plb_launch_gemm_tn)(C.byref(p), stream())
    assert rc == 0, rc
    rc = L.plb_launch_reduce_slabs(slab.data_ptr(), splits, N * K, out.data_ptr(), 0, stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out


def attn_args(qkv, lengths, B, S, NH):
    H = NH * 64
    dev = qkv.device
    p = _lib.PlbAttn()
    ctx = torch.zeros((B * S, H), dtype=torch.bfloat16, device=dev)
    lse = torch.zeros((B, NH, S), dtype=torch.float32, device=dev)
    p.qkv, p.ldqkv = qkv.data_ptr(), qkv.stride(0)
    p.lengths = lengths.data_ptr() if lengths is not None else None
    p.B, p.S, p.NH, p.H, p.scale = B, S, NH, H, 0.125
    p.ctx, p.ldctx, p.lse = ctx.data_ptr(), H, lse.data_ptr()
    return p, ctx, lse


def torch_attention(qkv, lengths, B, S, NH):
    """fp32 reference on the same (bf16-rounded) inputs; returns ctx [B*S,H], lse [B,NH,S], and a
    function computing dqkv for a given dctx."""
    H = NH * 64
    x = qkv.float().detach().clone().requires_grad_(True)
    q, k, v = [t.reshape(B, S, NH, 64).transpose(1, 2) for t in x.split(H, dim=1)]
    s = (q @ k.transpose(2, 3)) * 0.125
    if lengths is not None:
        keymask = torch.arange(S, device=qkv.device)[None, :] >= lengths[:, None].long()
        s = s.masked_fill(keymask[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B * S, H)

    def grad(dctx):
        (g,) = torch.autograd.grad(ctx, x, dctx.float())
        return g

    return ctx.detach(), lse.detach(), grad
