"""configs/config.yml's own batch on ONE GPU (-m gpu): batch_size 96 x seq 512 = 49,152 tokens
(/root/reference/configs/config.yml:16, modal_main.py:43 trains it on one A100). Round 1 refused any batch above
38,400 tokens in the embedding-gradient scatter. The reference comparison at this size is tests/golden/real_s512_b96.npz
(round 5, test_gpu_engine.py::test_real_model_against_reference_probes); here the step is checked through size-independent properties: the embedding gradients against a torch
index_add over the kernel's own per-token gradient rows, gradient sums that must vanish, and agreement of the
batch's loss with the mean of its three 32-sample thirds (every sample has masked phonemes, so the per-sample-mean
loss is linear in the samples)."""
import numpy as np
import pytest
import torch

from gpu_util import rel_l2
import plbert_amd
from plbert_amd.train import PLBertTrainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden,heads,inter", [(256, 4, 512), (768, 12, 2048)])
def test_96x512_step(hidden, heads, inter):
    """hidden 768: the fused LayerNorm launches have 384 row blocks x 2 column tiles = 768 workgroups on 256 CUs — three
    rounds, the one shape in the suite where the two members of a hand-off are not trivially co-resident from the start
    (forward progress rests on adjacent dispatch slots; the bounded wait's error word must stay 0)."""
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter,
                                  max_position_embeddings=512, num_hidden_layers=2)
    B, S = 96, 512
    labels, masked, lens, idx = plbert_amd.synthetic_batch(B, S, seed=31)
    tr = PLBertTrainer(cfg, 188, max_batch=B, max_seq=S, lr=1e-3, seed=2)
    eng = tr.engine
    full = tr.stage_batch(labels, masked, lens, idx)
    loss = float(tr.loss_and_grads(full).item())
    torch.cuda.synchronize()
    g = {k: eng.view(k, of=eng.grads).clone() for k in eng.layout}
    # (1) linearity over samples: count = B non-empty samples, so loss(B) = mean of the thirds' losses
    thirds = []
    for r in range(3):
        sl = slice(32 * r, 32 * r + 32)
        thirds.append(float(tr.loss_and_grads(tr.stage_batch(labels[sl], masked[sl], lens[sl], idx[sl])).item()))
    assert abs(loss - np.mean(thirds)) / loss < 2e-4
    # (2) embedding gradients = scatter-add of the per-token rows (dx is the kernel's own fp32 workspace image:
    #     recompute it by differencing is not possible from outside, so use the invariants of the scatter instead)
    tr.loss_and_grads(full)
    torch.cuda.synchronize()
    gw = eng.view("encoder.embeddings.word_embeddings.weight", of=eng.grads)
    gp = eng.view("encoder.embeddings.position_embeddings.weight", of=eng.grads)
    gt = eng.view("encoder.embeddings.token_type_embeddings.weight", of=eng.grads)
    assert float(gw[0].abs().max()) == 0.0                          # padding row: no gradient
    ids_present = np.unique(masked)
    absent = np.setdiff1d(np.arange(188), ids_present)
    assert float(gw[torch.from_numpy(absent).cuda()].abs().max()) == 0.0
    # every token's row lands in exactly one word row and one position row: the three tables see the same total
    assert rel_l2(gw.double().sum(0).float(), gp.double().sum(0).float()) < 1e-4
    assert rel_l2(gt[0], gp.double().sum(0).float()) < 1e-4 and float(gt[1].abs().max()) == 0.0
    # deterministic: a second evaluation is bit-identical
    assert all(torch.equal(g[k], eng.view(k, of=eng.grads)) for k in g)
    # (3) and it trains
    l1 = float(tr.step(full).item())
    for _ in range(3):
        l2 = float(tr.step(full).item())
    assert l2 < l1
    assert eng.status()["ln_exchange_timeouts"] == 0


def test_embed_scatter_matches_index_add_beyond_one_chunk():
    """Kernel level: T = 49,152 > the 32,768-token LDS chunk; dword/dpos against torch.index_add (fp32, order differs)."""
    import ctypes as C
    from plbert_amd import _lib
    from gpu_util import stream
    L = _lib.lib()
    T, S, E, V, P = 49152, 512, 128, 188, 512
    rs = np.random.RandomState(8)
    ids_np = rs.randint(1, V, size=T).astype(np.int64)
    ids_np[rs.rand(T) < 0.2] = 186                                  # a heavy row (the separator)
    ids_np[rs.rand(T) < 0.05] = 0                                   # padding: no gradient
    ids = torch.from_numpy(ids_np).cuda()
    dx = torch.randn(T, E, device="cuda")
    dword = torch.full((V, E), 7.0, device="cuda")
    dpos = torch.full((P, E), 7.0, device="cuda")
    p = _lib.PlbEmbed()
    p.ids, p.T, p.S, p.E, p.V = ids.data_ptr(), T, S, E, V
    p.dx, p.dword, p.dpos = dx.data_ptr(), dword.data_ptr(), dpos.data_ptr()
    assert L.plb_launch_embed_scatter(C.byref(p), P, stream()) == 0
    torch.cuda.synchronize()
    want_w = torch.zeros(V, E, device="cuda", dtype=torch.float64).index_add_(0, ids, dx.double())
    want_w[0] = 0
    want_p = dx.double().view(T // S, S, E).sum(0)
    assert rel_l2(dword, want_w.float()) < 1e-6
    assert rel_l2(dpos, want_p.float()) < 1e-6
