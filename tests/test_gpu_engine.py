"""End-to-end parity (-m gpu): the HIP engine through the C ABI against the oracle and the goldens
captured from the reference.  Tolerances are for a bf16-activation / fp32-accumulate path checked
against fp32 references; they are stated next to each assertion."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
from gpu_util import rel_l2
from oracle import albert_np as onp
import plbert_amd
from plbert_amd.engine import HipEngine

pytestmark = pytest.mark.gpu


def _batch(g):
    idx = [list(map(int, x)) for x in g["index"]]
    return g["labels"], g["masked"], [int(x) for x in g["lengths"]], idx


KEY_BIAS = "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.key.bias"
QUERY_BIAS = KEY_BIAS.replace("key", "query")


def _grad_close(eng, name, want, tol):
    """Per-tensor relative L2. The key bias is special: softmax is invariant to a per-query shift of the
    scores, so its true gradient is exactly 0 (the reference holds ~1e-12 of rounding noise); there we
    bound the noise against the query-bias gradient instead."""
    got = eng.view(name, of=eng.grads).cpu()
    if name == KEY_BIAS:
        scale = float(eng.view(QUERY_BIAS, of=eng.grads).double().norm())
        assert float(got.double().norm()) < 2e-2 * scale, (name, float(got.double().norm()), scale)
        return
    want = torch.as_tensor(want)
    assert got.shape == want.shape
    assert rel_l2(got, want) < tol, (name, rel_l2(got, want))


def _valid(g):
    return np.arange(g["labels"].shape[1])[None, :] < g["lengths"][:, None]


def _engine(g, max_batch=None, max_seq=None):
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    eng = HipEngine(pcfg, int(g["num_phonemes"]), int(g["num_tokens"]), max_batch=max_batch or B, max_seq=max_seq or S)
    eng.load_state_dict(sd)
    return eng, ocfg, pcfg, sd


def _step_inputs(g):
    labels, masked, lengths, idx = _batch(g)
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    return masked, labels, np.asarray(lengths, np.int32), off, flat, int(off[-1])


@pytest.mark.parametrize("name", ["small_h128", "small_h128_multitask"])
def test_forward_full_logits(name):
    g = load_golden(name)
    eng, ocfg, pcfg, sd = _engine(g)
    multitask = int(g["num_tokens"]) > 0
    hid, ph, tk = eng.forward(g["masked"], g["lengths"].astype(np.int32), want_hidden=True, want_token=multitask)
    torch.cuda.synchronize()
    v = _valid(g)
    # bf16 activations through 2 layers: hidden states are O(1) after LayerNorm
    assert np.abs(hid.cpu().numpy()[v] - g["hidden"][v]).max() < 6e-2
    assert rel_l2(torch.from_numpy(hid.cpu().numpy()[v]), torch.from_numpy(g["hidden"][v])) < 1e-2
    assert np.abs(ph.cpu().numpy()[v] - g["logits"][v]).max() < 3e-2   # logits <= ~3e-2 abs
    if multitask:
        assert np.abs(tk.cpu().numpy()[v] - g["token_logits"][v]).max() < 3e-2


def test_loss_and_grads_small():
    g = load_golden("small_h128")
    eng, ocfg, pcfg, sd = _engine(g)
    masked, labels, lens, off, flat, n = _step_inputs(g)
    loss = eng.loss_fwd_bwd(masked, labels, lens, off, flat, n)
    torch.cuda.synchronize()
    ref = float(g["loss"])
    assert abs(float(loss.item()) - ref) / ref < 1e-3          # north_star: loss within 1e-3 relative
    for k in g["grad_names"]:
        _grad_close(eng, str(k), g["grad/" + str(k)], 4e-2)  # bf16 backward, per-tensor relative L2
    # the pooler gets no gradient and lies outside the AdamW range
    off_pool = eng.layout["encoder.pooler.weight"][0]
    assert off_pool >= eng.trainable


def test_adamw_trajectory_small():
    g = load_golden("small_h128")
    eng, ocfg, pcfg, sd = _engine(g)
    masked, labels, lens, off, flat, n = _step_inputs(g)
    losses = []
    for step in range(1, len(g["losses"]) + 1):
        loss = eng.loss_fwd_bwd(masked, labels, lens, off, flat, n)
        losses.append(float(loss.item()))
        eng.adamw_step(step, lr=1e-3)
    torch.cuda.synchronize()
    assert np.allclose(losses, g["losses"], rtol=2e-3)
    # The optimizer arithmetic itself is pinned to 1e-6 against torch (tests/test_gpu_adamw_kernel.py); what is left
    # here is the bf16 backward's gradient noise seen through Adam's first steps, which move every weight by ~lr times
    # sign-like m / sqrt(v). EVERY tensor's parameter change is compared (relative L2; measured 0.02-0.10). The key bias
    # is excluded: its true gradient is 0 (softmax shift invariance), so Adam turns rounding noise into +-lr steps in
    # the reference as well as here.
    for k in g.files:
        if not k.startswith("final/") or k.endswith("attention.key.bias"):
            continue
        k = k[len("final/"):]
        d_ref = torch.from_numpy(g["final/" + k] - sd[k])
        if float(d_ref.norm()) == 0.0:
            continue
        d_got = eng.view(k).cpu() - torch.from_numpy(sd[k])
        assert rel_l2(d_got, d_ref) < 0.12, (k, rel_l2(d_got, d_ref))
    assert torch.equal(eng.view("encoder.pooler.weight").cpu(), torch.from_numpy(sd["encoder.pooler.weight"]))


@pytest.mark.parametrize("name,steps", [("real_s128_b8", 5), ("real_s512_b2_ragged", 2), ("real_h1024_s256_b4", 2),
                                        ("real_s512_b32", 5), ("real_h1024_s512_b16", 2), ("real_s512_b32_ragged", 2),
                                        ("real_s512_b96", 2)])
def test_real_model_against_reference_probes(name, steps):
    """configs/config.yml model (768/12): loss, probe logits, grad norms, loss trajectory. real_h1024_s256_b4 is BASELINE
    configs[3]'s architecture (1024 / 24 layers / 16 heads / FFN 4096) captured from the reference at 4 x 256 = 1024
    tokens, one row ragged: the first three fixtures have T = 1024, so the engine runs its fused LayerNorm epilogues — with two
    column tiles per row block at H = 768 and FOUR at H = 1024 (the exchange configs[3] runs at full size).
    real_s512_b32 / real_h1024_s512_b16: the sizes BASELINE.json configs[1] / configs[3] STATE (32 x 512 = 16,384 and
    16 x 512 = 8,192 tokens: 128 / 64 row blocks, the weight-gradient split rule at 196,608 stacked rows, 384 / 256
    attention items), captured from the reference on exactly bench.py's rank-0 inputs — reference initialisation seed 0,
    synthetic_batch(B, 512, seed=1234), AdamW lr 7e-5; bench.py's own first steps are compared with the same fixtures
    (its loss_parity entry). real_s512_b32_ragged: the same size with 32 RAGGED samples (lengths 64 .. 512, longest first as
    the collater sorts them, dataloader.py:276-297): attention tile skipping, padded rows and the pruned last application at
    16,384 rows against the reference. real_s512_b96: configs/config.yml's own batch_size (49,152 rows: 384 row blocks, 1,152
    attention items, 589,824 stacked rows in the weight-gradient GEMMs)."""
    g = load_golden(name)
    eng, ocfg, pcfg, sd = _engine(g)
    masked, labels, lens, off, flat, n = _step_inputs(g)
    _, ph, _ = eng.forward(masked, lens)
    ph = ph.cpu().numpy()
    pb, ps = g["probe_b"], g["probe_s"]
    assert np.abs(ph[pb, ps] - g["probe_logits"]).max() < 3e-2
    losses = []
    for step in range(1, steps + 1):
        loss = eng.loss_fwd_bwd(masked, labels, lens, off, flat, n)
        if step == 1:
            torch.cuda.synchronize()
            for k, ref in zip(g["grad_names"], g["grad_l2"]):
                if str(k) == KEY_BIAS:
                    continue  # exactly-zero gradient, see _grad_close
                got = float(eng.view(str(k), of=eng.grads).double().norm())
                assert abs(got - ref) <= 3e-2 * ref + 1e-6, (k, got, ref)
                flatg = eng.view(str(k), of=eng.grads).flatten().cpu().numpy()
                pv = g["gprobe_val/" + str(k)]
                assert np.abs(flatg[g["gprobe_idx/" + str(k)]] - pv).max() <= 0.1 * np.abs(flatg).max() + 1e-7, k
        losses.append(float(loss.item()))
        eng.adamw_step(step, lr=7e-5)
    assert np.allclose(losses, g["losses"][:steps], rtol=1e-3), (losses, g["losses"][:steps])
    assert eng.status()["ln_exchange_timeouts"] == 0


@pytest.mark.parametrize("name", ["real_s128_b8", "real_s512_b2_ragged", "real_h1024_s256_b4"])
def test_last_application_on_masked_rows_only_equals_the_full_evaluation(name):
    """A phoneme-only loss call evaluates what lies behind the attention of its LAST application on the masked rows alone
    (include/plbert.h: plb_last_application_rows): those are the only rows of that part the loss reads (train.py:107-131) and
    the only ones with a non-zero output gradient. Against the same call with every row evaluated (plb_set_prune_last(0)):
    the two are different bf16 evaluations of the same function (the compact part runs GEMM + LayerNorm kernels and gelu from
    the stored pre-activation where the full evaluation runs the fused epilogues), so they agree as two bf16 runs do: loss
    1e-4 (the reference bar is 1e-3), every gradient tensor 1.5e-2 relative L2 (against the reference: 4e-2); the AdamW
    trajectories stay within the 1e-3 the reference comparison allows — after an update they are two equally valid runs, not one: Adam turns the rounding
    noise of (near-)zero gradient coordinates into +-lr steps, differently in the two; the validation call (plb_loss_fwd)
    gives the training call's loss bit for bit in both modes."""
    from plbert_amd import _lib
    g = load_golden(name)
    L = _lib.lib()
    masked, labels, lens, off, flat, n = _step_inputs(g)
    out = {}
    try:
        for mode in (0, 1):
            L.plb_set_prune_last(mode)
            eng, ocfg, pcfg, sd = _engine(g)
            l_val = float(eng.loss_fwd(masked, labels, lens, off, flat, n).item())
            losses = []
            for step in range(1, 4):
                loss = eng.loss_fwd_bwd(masked, labels, lens, off, flat, n)
                if step == 1:
                    torch.cuda.synchronize()
                    grads = eng.grads[: eng.trainable].clone()
                    rows = eng.last_application_rows()
                losses.append(float(loss.item()))
                eng.adamw_step(step, lr=7e-5)
            assert l_val == losses[0]
            out[mode] = (losses, grads, rows, eng)
    finally:
        L.plb_set_prune_last(-1)
    (lf, gf, rf, ef), (lp, gp, rp, ep) = out[0], out[1]
    assert rf[0] == rf[1] and rp[0] < rp[1] // 2 + 1 and rp[0] % 128 == 0, (rf, rp)     # the second run really was pruned
    assert abs(lp[0] - lf[0]) <= 1e-4 * lf[0] and np.allclose(lp, lf, rtol=1e-3), (lp, lf)
    for k, (o, sz, shp) in ep.layout.items():
        if o + sz > ep.trainable or k == KEY_BIAS:
            continue
        a, b = gp[o:o + sz], gf[o:o + sz]
        assert rel_l2(a, b) < 1.5e-2, (k, rel_l2(a, b))
    assert abs(lp[0] - float(g["loss"])) / float(g["loss"]) < 1e-3


def test_against_oracle_random_shapes():
    """Oracle parity on shapes the goldens do not hold: ragged batch, S not a multiple of 64."""
    ocfg = onp.Config(embedding_size=128, hidden_size=256, num_attention_heads=4, intermediate_size=512,
                      num_hidden_layers=3)
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=128, hidden_size=256, num_attention_heads=4,
                                   intermediate_size=512, num_hidden_layers=3)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=5)
    rs = np.random.RandomState(3)
    B, S = 5, 90
    lengths = [90, 77, 64, 13, 1]
    labels = np.zeros((B, S), np.int64)
    masked = np.zeros((B, S), np.int64)
    idx = []
    for b, L in enumerate(lengths):
        labels[b, :L] = rs.randint(1, 185, size=L)
        masked[b, :L] = labels[b, :L]
        ii = sorted(rs.choice(L, size=max(1, L // 7), replace=False).tolist()) if b != 3 else []
        masked[b, ii] = 185
        idx.append(ii)
    loss_ref, pred_ref, G = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx)
    eng = HipEngine(pcfg, 188, 0, max_batch=8, max_seq=128)   # capacity larger than the batch
    eng.load_state_dict(sd)
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    loss = eng.loss_fwd_bwd(masked, labels, np.asarray(lengths, np.int32), off, flat, int(off[-1]))
    assert abs(float(loss.item()) - float(loss_ref)) / float(loss_ref) < 1e-3
    for k, want in G.items():
        _grad_close(eng, k, want, 4e-2)
    # a second, smaller batch through the same engine: stale padding rows must not leak into dW
    loss_ref2, _, G2 = onp.loss_and_grads(ocfg, sd, masked[:2, :70], labels[:2, :70], [70, 70], [idx[0][:3], [5]])
    off2, flat2 = plbert_amd.masked_indices_to_csr([idx[0][:3], [5]])
    loss2 = eng.loss_fwd_bwd(masked[:2, :70].copy(), labels[:2, :70].copy(), np.asarray([70, 70], np.int32), off2, flat2, int(off2[-1]))
    assert abs(float(loss2.item()) - float(loss_ref2)) / float(loss_ref2) < 1e-3
    for k, want in G2.items():
        _grad_close(eng, k, want, 4e-2)


@pytest.mark.parametrize("B", [4, 2])
def test_large_config_hidden1024_against_oracle(B, monkeypatch):
    """BASELINE.json configs[3] architecture (hidden 1024, 16 heads, FFN 4096 — heads/FFN assumed ALBERT-large as
    SURVEY.md §8 does) at reduced depth/batch so the fp32 oracle finishes in seconds. B = 4: 4 x 256 = 1024 tokens, the
    engine fuses LayerNorm into the GEMM epilogues with nbn = 4 column tiles per row block (H = 1024 = 4 x 256: the
    four-member exchange that configs[3] runs at full size), one row ragged; B = 2: 512 tokens, the unfused kernels —
    the control. Every gradient tensor against the oracle in both."""
    ocfg = onp.Config(embedding_size=128, hidden_size=1024, num_attention_heads=16, intermediate_size=4096,
                      num_hidden_layers=3)
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=128, hidden_size=1024, num_attention_heads=16,
                                   intermediate_size=4096, num_hidden_layers=3)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=11)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(B, 256, seed=77)
    lengths = [256] * (B - 1) + [201]
    idx = idx[:-1] + [[i for i in idx[-1] if i < 201]]
    labels[-1, 201:] = 0
    masked[-1, 201:] = 0
    loss_ref, pred_ref, G = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx)
    eng = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=256)
    eng.load_state_dict(sd)
    _, ph, _ = eng.forward(masked, np.asarray(lengths, np.int32))
    v = np.arange(256)[None, :] < np.asarray(lengths)[:, None]
    assert np.abs(ph.cpu().numpy()[v] - pred_ref[v]).max() < 3e-2
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    # the fused forms must actually be what runs at B = 4 (and not at B = 2): count the launches by class
    from plbert_amd import _lib
    _lib.profile_enable(True)
    try:
        loss = eng.loss_fwd_bwd(masked, labels, np.asarray(lengths, np.int32), off, flat, int(off[-1]))
        torch.cuda.synchronize()
        prof = _lib.profile_read()
    finally:
        _lib.profile_enable(False)
    fused = prof.get("gemm_nt_lnfwd", {}).get("launches", 0) + prof.get("gemm_nt_lnbwd", {}).get("launches", 0)
    # 3 layers: 2 forward + 2 backward fused launches per layer, minus the last LN2 backward (its output gradient comes from
    # the head) = 6 + 5; with the last application's post-attention part on the masked rows only (small-shape launches:
    # plb_last_application_rows) its 2 forward launches and its LayerNorm-1 backward are not fused ones: 4 + 4
    rows, of = eng.last_application_rows()
    assert rows < of                                          # ~13 % masked positions: the call was pruned
    assert fused == (4 + 4 if B == 4 else 0), prof.keys()
    assert abs(float(loss.item()) - float(loss_ref)) / float(loss_ref) < 1e-3
    for k, want in G.items():
        _grad_close(eng, k, want, 4e-2)
    assert eng.status()["ln_exchange_timeouts"] == 0


def test_zero_masked_indices_gives_zero_loss_and_grads():
    g = load_golden("small_h128")
    eng, *_ = _engine(g)
    masked, labels, lens, off, flat, n = _step_inputs(g)
    eng.grads.fill_(1.0)
    off0 = np.zeros_like(off)
    loss = eng.loss_fwd_bwd(masked, labels, lens, off0, flat[:0], 0)
    assert float(loss.item()) == 0.0
    assert float(eng.grads[: eng.trainable].abs().max()) == 0.0


@pytest.mark.parametrize("B,S,lengths", [(1, 1, [1]), (1, 512, [512]), (5, 33, [33, 32, 2, 1, 1]), (3, 65, [65, 64, 63])])
def test_edge_shapes_against_oracle(B, S, lengths):
    """Edges of the batch space: a single token, the maximum sequence length, samples of length 1 and 2 (a
    softmax over one key), lengths straddling the 64-key attention tile."""
    ocfg = onp.Config(embedding_size=64, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                      num_hidden_layers=2)
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                   intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=13)
    rs = np.random.RandomState(100 * B + S)
    labels = np.zeros((B, S), np.int64)
    masked = np.zeros((B, S), np.int64)
    idx = []
    for b, L in enumerate(lengths):
        labels[b, :L] = rs.randint(1, 185, size=L)
        masked[b, :L] = labels[b, :L]
        ii = sorted(rs.choice(L, size=max(1, L // 5), replace=False).tolist())
        masked[b, ii] = 185
        idx.append(ii)
    loss_ref, pred_ref, G = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx)
    eng = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    _, ph, _ = eng.forward(masked, np.asarray(lengths, np.int32))
    v = np.arange(S)[None, :] < np.asarray(lengths)[:, None]
    assert np.abs(ph.cpu().numpy()[v] - pred_ref[v]).max() < 3e-2
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    loss = eng.loss_fwd_bwd(masked, labels, np.asarray(lengths, np.int32), off, flat, int(off[-1]))
    assert abs(float(loss.item()) - float(loss_ref)) / float(loss_ref) < 1e-3
    # with one key per query the softmax is constant and the q/k gradients are exactly 0 in the oracle: compare
    # absolute errors against the largest gradient of the step where a tensor's own norm is (near) zero
    scale = max(float(np.sqrt((np.asarray(w, np.float64) ** 2).sum())) for w in G.values())
    for k, want in G.items():
        got = eng.view(k, of=eng.grads).cpu().double()
        want = torch.as_tensor(want).double()
        err = float((got - want).norm())
        assert err <= 5e-2 * float(want.norm()) + 1e-4 * scale, (k, err, float(want.norm()), scale)


def test_step_is_graph_capturable():
    """Nothing in the engine allocates or synchronises: loss + backward (both streams, fork/join events, memsets)
    can be captured into a hipGraph and replayed; replay reproduces the eager result bit for bit."""
    g = load_golden("small_h128")
    eng, *_ = _engine(g)
    masked, labels, lens, off, flat, n = _step_inputs(g)
    dev = eng.device
    args = [torch.as_tensor(masked).to(dev), torch.as_tensor(labels).to(dev), torch.as_tensor(lens).to(dev),
            torch.as_tensor(off).to(dev), torch.as_tensor(flat).to(dev), n]
    eng.loss_fwd_bwd(*args)
    torch.cuda.synchronize()
    loss_eager = float(eng._loss.item())
    grads_eager = eng.grads[: eng.trainable].clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.loss_fwd_bwd(*args)
    eng.grads.zero_()
    eng._loss.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert float(eng._loss.item()) == loss_eager
    assert torch.equal(eng.grads[: eng.trainable], grads_eager)


def test_training_memorises_a_fixed_batch():
    """End-to-end sanity of forward + backward + AdamW together: 150 steps on one fixed batch of random data
    (nothing to learn but the batch itself) drive the loss far below ln(vocabulary) with everything finite."""
    from plbert_amd.train import PLBertTrainer

    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                   intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    tr = PLBertTrainer(pcfg, 188, max_batch=4, max_seq=64, lr=2e-3, seed=3)
    batch = tr.stage_batch(*plbert_amd.synthetic_batch(4, 64, seed=5))
    losses = [float(tr.step(batch).item()) for _ in range(150)]
    assert np.isfinite(losses).all()
    assert losses[0] > 4.5 and losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert bool(torch.isfinite(tr.engine.params).all().item())


def test_full_size_properties():
    """BASELINE configs[1] at full size (768/12, batch 32 x 512), where the fp32 oracle would take minutes: the
    domain's size-independent properties. (1) two runs are bitwise identical; (2) the loss is the mean of the
    per-sample losses, so loss(batch) = mean of loss(quarter batches) and the gradients are the mean of the quarter
    gradients (linearity of the backward); (3) permuting the samples changes nothing but summation order."""
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                   max_position_embeddings=512, num_hidden_layers=12)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=17)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(32, 512, seed=4242)
    eng = HipEngine(pcfg, 188, 0, max_batch=32, max_seq=512)
    eng.load_state_dict(sd)

    def run(sel):
        off, flat = plbert_amd.masked_indices_to_csr([idx[i] for i in sel])
        loss = eng.loss_fwd_bwd(masked[sel], labels[sel], None, off, flat, int(off[-1]))
        torch.cuda.synchronize()
        return float(loss.item()), eng.grads[: eng.trainable].clone()

    full = list(range(32))
    l1, g1 = run(full)
    l2, g2 = run(full)
    assert l1 == l2 and torch.equal(g1, g2)                                   # (1) reproducible bit for bit
    assert abs(l1 - np.log(188)) < 0.5                                        # random weights: near-uniform predictions
    quarters = [run(list(range(q * 8, q * 8 + 8))) for q in range(4)]
    assert abs(l1 - np.mean([q[0] for q in quarters])) / l1 < 2e-4            # (2) loss is a mean over samples
    gq = torch.stack([q[1] for q in quarters]).mean(0)
    assert rel_l2(g1.cpu(), gq.cpu()) < 2e-2                                  # bf16 dY/X stashes, different row splits
    perm = np.random.RandomState(1).permutation(32).tolist()
    l3, g3 = run(perm)
    assert abs(l3 - l1) / l1 < 2e-5
    assert rel_l2(g3.cpu(), g1.cpu()) < 1e-2                                  # (3) order of summation only


def _size_independent_properties(pcfg, B, S, fp8=False, tol_lin=2e-2, tol_perm=1e-2):
    """Shared by the full-size tests: bitwise repeat, loss / gradient linearity over quarters of the batch, permutation
    of the samples. fp8: after the calibration call (which runs in bf16 and records the maxima) every call below uses the
    SAME delayed scales only if nothing in between updates them differently — each run() therefore restores the scaling
    state is not needed: the property checks compare calls whose scales come from the same preceding call sequence."""
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=17)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(B, S, seed=4242)
    eng = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S)
    eng.load_state_dict(sd)

    def run(sel):
        off, flat = plbert_amd.masked_indices_to_csr([idx[i] for i in sel])
        loss = eng.loss_fwd_bwd(masked[sel], labels[sel], None, off, flat, int(off[-1]))
        torch.cuda.synchronize()
        return float(loss.item()), eng.grads[: eng.trainable].clone()

    full = list(range(B))
    if fp8:
        eng.set_fp8(True)
        run(full)                                                             # calibration call (bf16 arithmetic)
        run(full)                                                             # first fp8 call: scales from the calibration
    l1, g1 = run(full)
    l2, g2 = run(full)
    if fp8:
        # delayed scaling: call n uses the maxima of call n-1; two consecutive calls on the same batch agree to the
        # change of the scales between them, not bit for bit
        assert abs(l1 - l2) / l1 < 2e-3 and rel_l2(g2.cpu(), g1.cpu()) < 5e-2
    else:
        assert l1 == l2 and torch.equal(g1, g2)                               # (1) reproducible bit for bit
    assert abs(l1 - np.log(188)) < 0.5                                        # random weights: near-uniform predictions
    assert bool(torch.isfinite(g1).all().item())
    q = B // 4
    quarters = [run(list(range(k * q, k * q + q))) for k in range(4)]
    assert abs(l1 - np.mean([x[0] for x in quarters])) / l1 < (2e-2 if fp8 else 2e-4)   # (2) loss is a mean over samples
    gq = torch.stack([x[1] for x in quarters]).mean(0)
    assert rel_l2(g1.cpu(), gq.cpu()) < tol_lin
    perm = np.random.RandomState(1).permutation(B).tolist()
    l3, g3 = run(perm)
    assert abs(l3 - l1) / l1 < (2e-2 if fp8 else 2e-5)
    assert rel_l2(g3.cpu(), g1.cpu()) < tol_perm                               # (3) order of summation only
    assert eng.status()["ln_exchange_timeouts"] == 0                          # every in-launch hand-off of the LayerNorm epilogues arrived
    return eng


def test_full_size_properties_config_d():
    """BASELINE configs[3] at full size: hidden 1024 / 24 shared layers / 16 heads / FFN 4096, batch 16 x 512 — the stash
    offsets at L = 24, the LayerNorm partial blocks, the weight-gradient split rule at Mtot = 196,608 x {1024, 3072,
    4096} and the 16-head attention grids only exist at this size (the golden-vector test of this architecture is 3
    layers of 2 x 256). ~13 GB of workspace."""
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=1024, num_attention_heads=16, intermediate_size=4096,
                                   max_position_embeddings=512, num_hidden_layers=24)
    eng = _size_independent_properties(pcfg, 16, 512)
    assert eng.ws_bytes > 8e9


def test_full_size_properties_fp8():
    """BASELINE configs[4] at the headline size (768/12, batch 32 x 512) in fp8 mode — since round 4 EVERY projection GEMM
    of the layer (forward, dX and the weight gradients) on e4m3 / e5m2 images: finite, near ln(188), linear over quarter
    batches and permutation-invariant within the fp8 path's stated tolerances. The comparisons here are between fp8
    evaluations whose delayed scales come from DIFFERENT previous calls (a quarter batch, whose gradients are 4x the full
    batch's, is quantised with the scales the full batch left, and vice versa). Measured (tools/fp8_diag.py): every call
    within 0.10-0.11 of the bf16 path in whole-gradient relative L2 whatever preceded it, a permuted batch within 0.025 of
    the unpermuted fp8 call. (Round 4 found a real bug with exactly this test: the scales were updated before the
    weight-gradient GEMMs had dequantised with them — 0.65 on the permuted call after the quarter batches.)"""
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                   max_position_embeddings=512, num_hidden_layers=12)
    _size_independent_properties(pcfg, 32, 512, fp8=True, tol_lin=0.25, tol_perm=0.08)


def test_layernorm_in_gemm_epilogue_matches_separate_kernels(monkeypatch):
    """The dense / FFN-output GEMMs carry their LayerNorm (forward) and the dX GEMMs the backward of the LayerNorm whose
    output gradient they produce (csrc/gemm_ln.hip; needs 1024-row multiples). Same step with PLBERT_LN_FUSE=off: the
    stored pre-LayerNorm sums are bit-identical by construction, the statistics are merged from per-tile partials instead
    of one two-pass sweep, so everything downstream agrees to bf16 rounding; both sit inside the reference tolerances
    (tests above run the fused path against the reference-captured fixture real_s512_b2_ragged: T = 1024)."""
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                   max_position_embeddings=512, num_hidden_layers=12)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=3)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(4, 512, seed=99)
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    out = {}
    for mode in ("off", "both"):
        monkeypatch.setenv("PLBERT_LN_FUSE", mode)
        eng = HipEngine(pcfg, 188, 0, max_batch=4, max_seq=512)
        eng.load_state_dict(sd)
        loss = eng.loss_fwd_bwd(masked, labels, None, off, flat, int(off[-1]))
        torch.cuda.synchronize()
        assert eng.status()["ln_exchange_timeouts"] == 0
        out[mode] = (float(loss.item()), eng.grads[: eng.trainable].clone().cpu(), dict(eng.layout), eng.trainable)
        del eng
    (l0, g0, layout, ntrain), (l1, g1, _, _) = out["off"], out["both"]
    assert abs(l0 - l1) / l0 < 2e-4
    assert rel_l2(g1, g0) < 1.5e-2
    for name, (o, n, shp) in layout.items():                                   # and tensor by tensor (LayerNorm affine, biases too)
        if o + n <= ntrain and float(g0[o:o + n].norm()) > 1e-6 and "key.bias" not in name:
            assert rel_l2(g1[o:o + n], g0[o:o + n]) < 3e-2, name
