"""N > 1 path of the PRODUCT on the GPU (-m gpu): two processes share the one MI355X of the test box, each runs
the HIP engine through PLBertTrainer on its shard of a global batch, and gradients are exchanged by the
trainer's own GradReducer (gloo carries the device tensors: RCCL refuses two ranks on one device). Must equal
the reference's DDP semantics computed with the oracle: per-rank local-count loss normalisation, mean of the
rank gradients, identical AdamW update everywhere."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _setup():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import plbert_amd
    from oracle import albert_np as onp

    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                   intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    ocfg = onp.Config(embedding_size=64, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                      num_hidden_layers=2)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=9)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(4, 48, seed=3)
    idx[1] = []            # the ranks' local counts of non-empty samples differ: 1 on rank 0, 2 on rank 1
    masked[1] = labels[1]
    return plbert_amd, onp, pcfg, ocfg, sd, (labels, masked, lengths, idx)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plbert_amd, onp, pcfg, ocfg, sd, batch = _setup()
    from plbert_amd.dist import shard_batch
    from plbert_amd.train import PLBertTrainer

    torch.cuda.set_device(0)
    start = {k: (v + 1.0 if rank == 1 else v) for k, v in sd.items()}    # rank 1 starts from garbage: broadcast fixes it
    tr = PLBertTrainer(pcfg, 188, max_batch=2, max_seq=48, lr=1e-3, state_dict=start)
    assert tr.world == world and tr.reducer.active
    lab, msk, lens, idx = shard_batch(batch, rank, world)
    loss = tr.step(tr.stage_batch(lab, msk, lens, idx))
    torch.cuda.synchronize()
    assert tr.engine.status()["ln_exchange_timeouts"] == 0
    out[rank] = (float(loss.item()), {k: v.cpu().numpy() for k, v in tr.engine.state_dict().items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_ddp_semantics():
    world = 2
    port = 29600 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    plbert_amd, onp, pcfg, ocfg, sd, batch = _setup()
    from plbert_amd.dist import shard_batch

    losses, grads = [], []
    for r in range(world):
        lab, msk, lens, idx = shard_batch(batch, r, world)
        loss, _, G = onp.loss_and_grads(ocfg, sd, msk, lab, lens, idx)
        losses.append(float(loss))
        grads.append(G)
    mean_g = {k: (grads[0][k] + grads[1][k]) / world for k in grads[0]}
    P = {k: v.astype(np.float32).copy() for k, v in sd.items()}
    onp.AdamW(lr=1e-3).step(P, mean_g)
    for r in range(world):
        assert abs(res[r][0] - losses[r]) / losses[r] < 1e-3                 # each rank reports its LOCAL loss
    for k in P:
        a, b = res[0][1][k], res[1][1][k]
        assert np.array_equal(a, b), k                                        # replicas stay bit-identical
    # Adam's first step moves every weight by ~lr in the direction of sign(g): compare the update in aggregate
    for k in ("encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn.weight", "phoneme_predictor.weight",
              "encoder.embeddings.word_embeddings.weight"):
        d_got = res[0][1][k] - sd[k]
        d_ref = P[k] - sd[k]
        rel = np.linalg.norm(d_got - d_ref) / np.linalg.norm(d_ref)
        assert rel < 0.15, (k, rel)
    assert np.array_equal(res[0][1]["encoder.pooler.weight"], sd["encoder.pooler.weight"])
