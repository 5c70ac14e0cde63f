"""The data-parallel exchange behind the C ABI (-m gpu): plb_comm_init / plb_allreduce_grads / plb_broadcast_params
with a real RCCL communicator. The test box has one GPU, so the communicator has ONE rank (RCCL refuses two ranks on a
device): what is pinned here is the protocol — unique id over the store, the piecewise all-reduce issued inside
plb_loss_fwd_bwd on the communication stream and joined before AdamW, the serial form, error paths — and that none of
it changes a single bit of the step. N > 1 arithmetic is pinned by tests/test_dist_gloo.py (CPU, world 2) and
tests/test_gpu_dist_two_ranks.py (two processes on this GPU over gloo); `bench.py --gpus 2` is run end to end below."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

import plbert_amd
from plbert_amd.engine import HipEngine
from plbert_amd.train import PLBertTrainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _cfg():
    return plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=256, num_attention_heads=4,
                                   intermediate_size=512, num_hidden_layers=3, max_position_embeddings=512)


def _run(comm, overlap, steps=3, num_tokens=0):
    cfg = _cfg()
    labels, masked, lens, idx = plbert_amd.synthetic_batch(4, 128, seed=21)
    tok = np.random.RandomState(2).randint(0, max(num_tokens, 1), size=(4, 128)).astype(np.int64) if num_tokens else None
    tr = PLBertTrainer(cfg, 188, max_batch=4, max_seq=128, lr=1e-3, seed=4, num_tokens=num_tokens,
                       force_collectives=comm != "none", comm=comm if comm != "none" else "auto", overlap=overlap)
    batch = tr.stage_batch(labels, masked, lens, idx, token_ids=tok)
    losses = [float(tr.step(batch).item()) for _ in range(steps)]
    torch.cuda.synchronize()
    out = (losses, tr.engine.params.clone(), tr.engine.grads.clone(), tr)
    return out


@pytest.fixture(scope="module")
def one_rank_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 1000))
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("num_tokens", [0, 512])
def test_rccl_exchange_is_bit_transparent(one_rank_group, num_tokens):
    base_l, base_p, base_g, tr0 = _run("none", True, num_tokens=num_tokens)
    assert tr0.comm == "none"
    for overlap in (True, False):
        l, p, g, tr = _run("rccl", overlap, num_tokens=num_tokens)
        assert tr.comm == "rccl"
        rank, world, ver = tr.engine.comm_info()
        assert (rank, world) == (0, 1) and ver >= 20000
        assert l == base_l and torch.equal(p, base_p) and torch.equal(g, base_g), overlap
        tr.engine.comm_destroy()


def test_allreduce_counts_ranks_and_zero_loss_batches(one_rank_group):
    l, p, g, tr = _run("rccl", True, steps=1)
    eng = tr.engine
    eng.grads.fill_(1.0)
    eng.set_grad_overlap(False)
    # a loss call resets the "already reduced" state; the explicit call then reduces once
    labels, masked, lens, idx = plbert_amd.synthetic_batch(4, 128, seed=22)
    empty = tr.stage_batch(labels, labels, lens, [[] for _ in idx])
    loss = tr.step(empty)                        # n_masked == 0 at world 1 under force: gradients zero, still exchanged
    torch.cuda.synchronize()
    assert float(loss.item()) == 0.0 and float(eng.grads[: eng.trainable].abs().max()) == 0.0
    eng.set_grad_overlap(True)
    loss = tr.step(empty)
    torch.cuda.synchronize()
    assert float(loss.item()) == 0.0
    # second communicator on the same engine is refused; after destroy a new one works
    with pytest.raises(RuntimeError):
        eng.comm_init(HipEngine.comm_unique_id(), 0, 1)
    eng.comm_destroy()
    eng.comm_init(HipEngine.comm_unique_id(), 0, 1)
    eng.broadcast_params(0)
    torch.cuda.synchronize()
    eng.comm_destroy()


def _bench(*extra):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--seq", "128", "--no-cpu-baseline", "--no-traffic", *extra], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_gpus_2_as_typed_self_launches_its_ranks():
    """`python bench.py --gpus 2` from a plain shell: the parent spawns the ranks before any GPU call and relays
    rank 0's line. On this one-GPU box the two ranks share the device and exchange over gloo (the line says so)."""
    out = _bench("--gpus", "2", "--no-roofline")
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["comm"]["mode"] == "torch" and out["comm"]["shared_devices"] is True
    assert out["value"] > 0 and np.isfinite(out["step_loss"])


def test_bench_rccl_rehearsal_at_world_1_reports_exchange_cost():
    out = _bench("--force-dist", "--no-roofline")
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1
    c = out["comm"]
    assert c["mode"] == "rccl" and c["rccl_version"] >= 20000
    for k in ("allreduce_ms_per_step_serial", "step_ms_overlap", "step_ms_serial", "step_ms_no_exchange",
              "allreduce_ms_per_step_exposed"):
        assert k in c and np.isfinite(c[k])
    assert out["staged"]["value"] > 0
