"""N > 1 path on CPU: two gloo ranks run the product's sharding + gradient exchange (plbert_amd.dist)
with the oracle standing in for the HIP step, and must reproduce what the reference's DDP does —
mean over ranks of per-rank gradients, each rank normalising by its LOCAL count of non-empty samples
(train.py:129) — followed by identical AdamW updates on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import plbert_amd
    from oracle import albert_np as onp

    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                   intermediate_size=256, num_hidden_layers=2)
    ocfg = onp.Config(embedding_size=64, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                      num_hidden_layers=2)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=9)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(4, 24, seed=3)
    idx[1] = []            # one empty sample: the ranks' local counts differ (2 vs 1 non-empty)... rank 0 has [s0, s1]
    masked[1] = labels[1]
    return plbert_amd, onp, pcfg, ocfg, sd, (labels, masked, lengths, idx)


def _flatten(G, names):
    return np.concatenate([G[n].reshape(-1) for n in names])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plbert_amd, onp, pcfg, ocfg, sd, batch = _setup()
    from plbert_amd.dist import GradReducer, shard_batch

    names = [n for n in plbert_amd.param_shapes(pcfg, 188) if "pooler" not in n]
    red = GradReducer()
    assert (red.rank, red.world) == (rank, world)
    # start-up broadcast: rank 1 starts from garbage and must end with rank 0's parameters
    flat_p = torch.from_numpy(_flatten(sd, names).copy())
    if rank == 1:
        flat_p += 1.0
    red.broadcast_(flat_p)
    assert np.array_equal(flat_p.numpy(), _flatten(sd, names))
    lab, msk, lens, idx = shard_batch(batch, rank, world)
    loss, _, G = onp.loss_and_grads(ocfg, sd, msk, lab, lens, idx)
    flat_g = torch.from_numpy(_flatten(G, names).astype(np.float32))
    third = flat_g.numel() // 3
    red.all_reduce_(flat_g, pieces=[(0, third), (third, 2 * third), (2 * third, flat_g.numel())])
    out[rank] = (float(loss), (flat_g / world).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_matches_ddp_semantics():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    plbert_amd, onp, pcfg, ocfg, sd, batch = _setup()
    from plbert_amd.dist import shard_batch

    names = [n for n in plbert_amd.param_shapes(pcfg, 188) if "pooler" not in n]
    per_rank = []
    for r in range(world):
        lab, msk, lens, idx = shard_batch(batch, r, world)
        loss, _, G = onp.loss_and_grads(ocfg, sd, msk, lab, lens, idx)
        per_rank.append((float(loss), _flatten(G, names)))
    want = (per_rank[0][1] + per_rank[1][1]) / world
    for r in range(world):
        assert abs(res[r][0] - per_rank[r][0]) < 1e-6            # each rank logs its LOCAL loss
        assert np.allclose(res[r][1], want, rtol=1e-5, atol=1e-8)  # identical averaged gradient on every rank
    # and this differs from single-process training on the global batch (count 3 vs local 2 and 1)
    _, _, Gfull = onp.loss_and_grads(ocfg, sd, batch[1], batch[0], batch[2], batch[3])
    assert not np.allclose(want, _flatten(Gfull, names), rtol=1e-3, atol=1e-7)


def test_shard_batch_edges():
    from plbert_amd.dist import shard_batch

    lab = np.arange(24).reshape(6, 4)
    b = (lab, lab + 100, [4, 4, 3, 3, 2, 1], [[0], [1], [], [2], [0], [0]])
    s1 = shard_batch(b, 1, 3)
    assert s1[0].tolist() == lab[2:4].tolist() and s1[2] == [3, 3] and s1[3] == [[], [2]]
    with pytest.raises(ValueError):
        shard_batch(b, 0, 4)


def test_shard_batch_short_last_batch_follows_accelerate_even_batches():
    """The reference prepares its loaders with Accelerator(split_batches=True) (train.py:218-221) and accelerate's default
    even_batches=True: BatchSamplerShard completes a short last batch (drop_last=False: validation) to the FULL batch
    size with the indices of the pass's first batch, and every rank takes batch_size // world of it. shard_batch with
    the pass context must hand each rank exactly those samples (checked against accelerate's own sampler)."""
    from accelerate.data_loader import BatchSamplerShard
    from torch.utils.data import BatchSampler, SequentialSampler

    from plbert_amd.dist import shard_batch

    n, bs, world = 10, 4, 2
    lab = np.arange(n * 3).reshape(n, 3)
    lengths = [3] * n
    idx = [[i % 3] for i in range(n)]
    batches = [list(range(i, min(i + bs, n))) for i in range(0, n, bs)]          # what the un-sharded loader collates
    collate = lambda ids: (lab[ids], lab[ids] + 1000, [lengths[i] for i in ids], [idx[i] for i in ids])
    first = collate(batches[0])
    for rank in range(world):
        acc = list(BatchSamplerShard(BatchSampler(SequentialSampler(range(n)), bs, drop_last=False), num_processes=world,
                                     process_index=rank, split_batches=True, even_batches=True))
        assert len(acc) == len(batches)
        for ids, want in zip(batches, acc):
            got = shard_batch(collate(ids), rank, world, pad=True, batch_size=bs, first_batch=first)
            assert got[0][:, 0].tolist() == [3 * i for i in want], (rank, ids, want)
            assert got[3] == [idx[i] for i in want]
