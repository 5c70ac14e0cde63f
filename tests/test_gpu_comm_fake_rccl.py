"""The engine's OWN gradient exchange at world size 2 (-m gpu). RCCL refuses two ranks on one device and the test box has
one GPU, so the piecewise protocol of csrc/engine.cpp (pieces issued inside plb_loss_fwd_bwd on the communication
stream, the zero-masked rank that must issue the identical collective sequence, the dual-head token piece, the coverage
check of pieces_done) had only run where ncclAllReduce is the identity. tests/fake_rccl.cpp is a stand-in library with
RCCL's seven entry points — stream-ordered collectives over POSIX shared memory between processes that share the GPU —
loaded through PLBERT_RCCL_LIB. Checked against the same two ranks exchanging through torch.distributed / gloo
(dist.GradReducer, pinned to DDP's semantics by tests/test_gpu_dist_two_ranks.py): replicas must be bit-identical.
Reference: the DDP bucketed all-reduce the reference gets from accelerate (train.py:218-221, 356)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _setup(num_tokens, real=False):
    """real: the model and per-rank shape class of the bench (768 / 12 layers / 12 heads / FFN 2048, 2 x 512 = 1024 tokens
    per rank): the launches between the pieces are then the fused LayerNorm + gelu-stash forms and the 256x256 token-major
    weight-gradient GEMMs the headline step runs, not the small-shape fallbacks of the 256-wide toy model."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import plbert_amd

    if real:
        cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                      num_hidden_layers=12, max_position_embeddings=512)
        S = 512
    else:
        cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=256, num_attention_heads=4,
                                      intermediate_size=512, num_hidden_layers=3, max_position_embeddings=512)
        S = 64
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(4, S, seed=3)
    tok = np.random.RandomState(5).randint(0, max(num_tokens, 1), size=(4, S)).astype(np.int64) if num_tokens else None
    return plbert_amd, cfg, (labels, masked, lengths, idx), tok


def _worker(rank, world, port, lib, num_tokens, empty_rank, real, fp8, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PLBERT_RCCL_LIB=lib, FAKE_RCCL_TIMEOUT_S="60")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plbert_amd, cfg, batch, tok = _setup(num_tokens, real)
    S = batch[0].shape[1]
    from plbert_amd.dist import shard_batch
    from plbert_amd.train import PLBertTrainer

    torch.cuda.set_device(0)
    labels, masked, lengths, idx = batch
    if empty_rank is not None:  # this rank's shard holds no masked phoneme: zero loss, zero gradients, SAME collectives
        per = len(lengths) // world
        for b in range(empty_rank * per, (empty_rank + 1) * per):
            idx[b] = []
            masked[b] = labels[b]
    lab, msk, lens, ix = shard_batch((labels, masked, lengths, idx), rank, world)
    tk = tok[rank * 2:(rank + 1) * 2] if tok is not None else None
    res = {}
    for mode, overlap in (("torch", True), ("rccl", True), ("rccl", False)):
        tr = PLBertTrainer(cfg, 188, max_batch=2, max_seq=S, lr=1e-3, seed=11, num_tokens=num_tokens, comm=mode,
                           overlap=overlap)
        assert tr.world == world and tr.comm == mode
        if mode == "rccl":
            assert tr.engine.comm_info() == (rank, world, 29999)          # the stand-in, not a real RCCL
        if fp8:
            tr.engine.set_fp8(True)                                         # step 1 calibrates (bf16), steps 2-3 run on the images
        # happens-before audit: every event record / stream wait of the steps below is mirrored in a
        # vector-clock model and every cross-stream buffer access checked against it; a violation fails the loss call
        tr.engine.hb_audit(True)
        b = tr.stage_batch(lab, msk, lens, ix, token_ids=tk)
        losses = [float(tr.step(b).item()) for _ in range(3)]
        torch.cuda.synchronize()
        rep = tr.engine.hb_report()
        assert rep["violations"] == 0, rep
        if empty_rank != rank:   # (a rank without masked phonemes has no backward: nothing crosses a stream)
            assert rep["checks"] > (50 if mode == "rccl" else 10), rep
        assert tr.engine.status()["ln_exchange_timeouts"] == 0              # two processes share the GPU: hand-offs still arrive
        pieces = tr.engine.comm_pieces() if mode == "rccl" else None
        if fp8:
            assert tr.engine.fp8_state() == (True, True)
        res[(mode, overlap)] = (losses, tr.engine.params.cpu().numpy().copy(), pieces)
        if mode == "rccl":
            tr.engine.comm_destroy()
        del tr
    L = C.CDLL(lib)
    L.fake_rccl_errors.restype = C.c_uint
    out[rank] = (res, int(L.fake_rccl_errors()))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(lib, num_tokens=0, empty_rank=None, real=False, fp8=False):
    port = 29800 + (os.getpid() % 1500)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, lib, num_tokens, empty_rank, real, fp8, out), nprocs=2, join=True)
        return dict(out)


def _check(res, expect_pieces_overlap):
    for r in (0, 1):
        assert res[r][1] == 0, "fake_rccl saw mismatched collectives or a time-out"
    ref0, ref1 = res[0][0][("torch", True)], res[1][0][("torch", True)]
    assert np.array_equal(ref0[1], ref1[1])                                  # gloo replicas agree (DDP semantics pinned elsewhere)
    for key in (("rccl", True), ("rccl", False)):
        for r in (0, 1):
            losses, params, pieces = res[r][0][key]
            assert losses == res[r][0][("torch", True)][0], (key, r)         # local losses: the exchange does not touch them
            assert np.array_equal(params, ref0[1]), (key, r)                 # bit-identical to the gloo exchange, on both ranks
            assert pieces[0] == (expect_pieces_overlap if key[1] else expect_pieces_overlap - 9), (key, pieces)


def test_piecewise_exchange_world2_matches_gloo(fake_lib):
    """Overlapped (10 pieces inside plb_loss_fwd_bwd) and serial (one all-reduce in plb_allreduce_grads) forms."""
    _check(_run_world2(fake_lib), expect_pieces_overlap=10)


def test_piecewise_exchange_world2_on_the_bench_model(fake_lib):
    """The same on the model and launch forms the bench runs (768 / 12, 1024 tokens per rank: LayerNorm in the GEMM epilogues,
    gelu-derivative stash, big-tile weight-gradient GEMMs with the pieces issued between them)."""
    _check(_run_world2(fake_lib, real=True), expect_pieces_overlap=10)


def test_piecewise_exchange_world2_fp8_calls(fake_lib):
    """fp8 mode on the bench model: the pieces travel between the fp8 weight-gradient GEMMs (12 x 1024 stacked rows), each
    rank quantises under its OWN delayed scales (the maxima are local; the weights' are identical), and the scale update
    at the end of the call must not disturb the exchange — the replicas stay bit-identical to the gloo exchange of the
    same fp8 steps, in the overlapped and in the serial form."""
    _check(_run_world2(fake_lib, real=True, fp8=True), expect_pieces_overlap=10)


def test_zero_masked_rank_issues_the_same_collectives(fake_lib):
    """Rank 1's shard has no masked phoneme: its loss call takes the early-return path (train.py:129) and must replay the
    10 ranges in the order a regular step issues them — a different sequence would deadlock or, worse, sum mismatched
    ranges; the stand-in checks (kind, count) of every collective across the ranks."""
    res = _run_world2(fake_lib, empty_rank=1)
    _check(res, expect_pieces_overlap=10)
    assert res[1][0][("rccl", True)][0] == [0.0, 0.0, 0.0]


def test_dual_head_token_piece_world2(fake_lib):
    """Dual-head step: the token head's gradients travel as an eleventh piece (two all-reduces in the serial form)."""
    _check(_run_world2(fake_lib, num_tokens=512), expect_pieces_overlap=11)


def test_forgotten_piece_is_caught(fake_lib):
    """A piece left out of the exchange (plb_debug_skip_piece) makes the loss call FAIL in pieces_done(): at world 1 a
    forgotten tensor would otherwise go unnoticed for ever (the all-reduce is the identity there)."""
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='{29900 + os.getpid() % 90}', PLBERT_RCCL_LIB={fake_lib!r})
import torch, torch.distributed as dist
import plbert_amd
from plbert_amd import _lib
from plbert_amd.train import PLBertTrainer
dist.init_process_group('gloo', rank=0, world_size=1)
cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=256, num_attention_heads=4,
                              intermediate_size=512, num_hidden_layers=2, max_position_embeddings=512)
labels, masked, lens, idx = plbert_amd.synthetic_batch(2, 64, seed=3)
tr = PLBertTrainer(cfg, 188, max_batch=2, max_seq=64, lr=1e-3, seed=1, force_collectives=True, comm='rccl', overlap=True)
b = tr.stage_batch(labels, masked, lens, idx)
tr.step(b); torch.cuda.synchronize()
assert tr.engine.comm_pieces()[0] == 10
L = _lib.lib()
L.plb_debug_skip_piece.argtypes = [__import__('ctypes').c_int]
L.plb_debug_skip_piece(3)
try:
    tr.step(b)
except RuntimeError as ex:
    assert 'gradient exchange covered' in str(ex), ex
    print('CAUGHT')
else:
    print('MISSED')
"""
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "CAUGHT" in r.stdout, r.stdout[-2000:]


# ---- the step's health word at world 2 -----------------------------------------------------------------------------------
def _fault_worker(rank, world, port, lib, out):
    """A fused LayerNorm hand-off times out on RANK 1 ONLY (plb_debug_ln_fault, process-wide in that rank's process). Its
    gradients are invalid and are summed into both replicas by the exchange: the word must travel with them, so that BOTH
    ranks return a NaN loss, leave the update out, raise HandoffTimeout from their next call and — after the retry — end
    bit-identical to an undisturbed run."""
    import math
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PLBERT_RCCL_LIB=lib, FAKE_RCCL_TIMEOUT_S="60")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import plbert_amd
    from plbert_amd import _lib
    from plbert_amd.dist import shard_batch
    from plbert_amd.engine import HandoffTimeout
    from plbert_amd.train import PLBertTrainer

    torch.cuda.set_device(0)
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                  max_position_embeddings=512, num_hidden_layers=2)
    lab, msk, lens, ix = shard_batch(plbert_amd.synthetic_batch(4, 512, seed=5), rank, world)
    L = _lib.lib()
    res = {}
    for mode, overlap in (("rccl", True), ("rccl", False), ("torch", True)):
        def make():
            tr = PLBertTrainer(cfg, 188, max_batch=2, max_seq=512, lr=1e-3, seed=3, comm=mode, overlap=overlap)
            return tr, tr.stage_batch(lab, msk, lens, ix)

        def drop(tr):
            if mode == "rccl":
                tr.engine.comm_destroy()

        tr, b = make()
        clean = [float(tr.step(b).item()) for _ in range(3)]
        torch.cuda.synchronize()
        clean_params = tr.engine.params.clone()
        drop(tr)
        del tr
        tr, b = make()
        first = float(tr.step(b).item())
        torch.cuda.synchronize()
        p1 = tr.engine.params.clone()
        if rank == 1:
            L.plb_debug_ln_fault(1, 1)
        try:
            bad = tr.step(b)
            torch.cuda.synchronize()
        finally:
            L.plb_debug_ln_fault(0, 0)
        r = dict(nan=math.isnan(float(bad.item())), untouched=bool(torch.equal(tr.engine.params, p1)), raised=False, skipped=None)
        try:
            tr.step(b)
        except HandoffTimeout as ex:
            r.update(raised=True, skipped=ex.skipped_updates)
        r["step_count"] = tr.step_count
        rest = [float(tr.step(b).item()) for _ in range(2)]
        torch.cuda.synchronize()
        r.update(losses_match=[first] + rest == clean, equals_clean=bool(torch.equal(tr.engine.params, clean_params)),
                 params=tr.engine.params.cpu().numpy().copy(), status=tr.engine.status())
        res[(mode, overlap)] = r
        drop(tr)
        del tr
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_handoff_timeout_on_one_rank_is_every_ranks(fake_lib):
    port = 29800 + ((os.getpid() + 700) % 1500)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fault_worker, args=(2, port, fake_lib, out), nprocs=2, join=True)
        res = dict(out)
    for key in (("rccl", True), ("rccl", False), ("torch", True)):
        for r in (0, 1):
            v = res[r][key]
            assert v["nan"], (key, r, "the failed step's loss must be NaN on every rank")
            assert v["untouched"], (key, r, "no rank may apply the update built from the poisoned sum")
            assert v["raised"] and v["skipped"] == 1 and v["step_count"] == 1, (key, r, v["raised"], v["skipped"], v["step_count"])
            assert v["losses_match"] and v["equals_clean"], (key, r)
            assert v["status"] == {"ln_exchange_timeouts": 0, "skipped_updates": 0}, (key, r, v["status"])
        assert np.array_equal(res[0][key]["params"], res[1][key]["params"]), key


@pytest.mark.parametrize("forgotten", [0, 2, 3, 5, 11, 14])
def test_the_audit_sees_a_forgotten_wait(fake_lib, forgotten):
    """The happens-before audit must FIND a missing hipStreamWaitEvent: the model forgets the n-th wait of a step (the HIP call
    is still made, so nothing races for real) — 0: the head piece behind the head's weight gradient, 2: the side stream
    behind the layer loop, 3: the Q/K/V piece behind its weight-gradient GEMM, 5: the first of the five small pieces behind the
    side stream's own event (the other four wait for the same point of that stream: implied by the first), 11: the last piece (dense.weight) behind its GEMM, 14: AdamW behind the communication stream — and
    the step (14: the next one) must fail, naming it. (12, the main stream's join of the side stream at the end of the tail, is
    implied in the overlapped form — AdamW waits for the communication stream, which has waited for the side stream's last
    event — and is what orders the two in the serial form.)"""
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='{29950 + (os.getpid() + forgotten) % 40}', PLBERT_RCCL_LIB={fake_lib!r})
import torch, torch.distributed as dist
import plbert_amd
from plbert_amd.train import PLBertTrainer
dist.init_process_group('gloo', rank=0, world_size=1)
cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=256, num_attention_heads=4,
                              intermediate_size=512, num_hidden_layers=2, max_position_embeddings=512)
labels, masked, lens, idx = plbert_amd.synthetic_batch(2, 64, seed=3)
tr = PLBertTrainer(cfg, 188, max_batch=2, max_seq=64, lr=1e-3, seed=1, force_collectives=True, comm='rccl', overlap=True)
b = tr.stage_batch(labels, masked, lens, idx)
tr.engine.hb_audit(True)
tr.step(b); tr.step(b); torch.cuda.synchronize()
rep = tr.engine.hb_report()
assert rep['violations'] == 0 and rep['checks'] > 50, rep
tr.engine.hb_audit(True, {forgotten})
try:
    tr.step(b); tr.step(b)
except RuntimeError as ex:
    assert 'happens-before audit' in str(ex), ex
    print('CAUGHT', ex)
else:
    print('MISSED', tr.engine.hb_report())
"""
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "CAUGHT" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_at_world_2_through_the_engines_own_exchange(fake_lib, ranks):
    """`python bench.py --gpus N --comm rccl` (N = 2, 4: ranks sharing the one GPU) with the stand-in library: the rehearsal of the driver's N > 1 run that one GPU
    allows — bench.py's own launcher, two real processes, the unique id travelling over the control plane, plb_comm_init at
    world 2, the overlapped-vs-serial calibration agreed between the ranks, the piece trace, the exposed-exchange figures and
    ONE JSON line from rank 0 (the 8-GPU scaling run issues exactly this sequence with librccl in the stand-in's place)."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, PLBERT_RCCL_LIB=fake_lib, FAKE_RCCL_TIMEOUT_S="120")
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--comm", "rccl", "--steps", "4",
                        "--warmup", "2", "--batch", "4", "--seq", "128", "--no-cpu-baseline", "--no-traffic", "--no-roofline"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    c = out["comm"]
    assert out["n_gpus"] == ranks and out["ranks_seen"] == ranks and out["config"]["global_batch"] == 4 * ranks
    assert c["mode"] == "rccl" and c["world"] == ranks, c
    assert c["pieces_per_step"] in (1, 10), c                         # the form the calibration kept: serial or overlapped
    tr = c["piece_trace"]
    assert len(tr["pieces"]) == 10 and all(p["done_ms"] >= p["released_ms"] for p in tr["pieces"]), tr
    for k in ("step_ms_overlap", "step_ms_serial", "step_ms_no_exchange", "allreduce_ms_per_step_serial"):
        assert k in c and np.isfinite(c[k]), (k, c)
    assert out["value"] > 0 and np.isfinite(out["step_loss"]) and out["ln_exchange_timeouts"] == 0
