"""Pin the numpy oracle to outputs of the reference itself (fixtures from oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import golden_cfg, load_golden
from oracle import albert_np as onp


def _batch(g):
    idx = [list(map(int, x)) for x in g["index"]]
    return g["labels"], g["masked"], [int(x) for x in g["lengths"]], idx


def _valid(g):
    L = g["lengths"]
    S = g["labels"].shape[1]
    return np.arange(S)[None, :] < L[:, None]


@pytest.mark.parametrize("name", ["tiny_h64", "small_h128"])
def test_full_tensors_forward_loss_grads(name):
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    labels, masked, lengths, idx = _batch(g)
    loss, pred, G = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx, dtype=np.float64)
    v = _valid(g)
    # padded query rows are garbage-but-finite in the reference; compare valid rows only
    assert np.abs(pred[v] - g["logits"][v]).max() < 1e-5
    assert abs(loss - float(g["loss"])) / float(g["loss"]) < 1e-6
    for k in g["grad_names"]:
        ref = g["grad/" + k]
        assert G[k].shape == ref.shape
        assert np.abs(G[k] - ref).max() < 2e-6 + 1e-4 * np.abs(ref).max(), k
    # parameters the reference leaves without a gradient (the pooler)
    for k in g["grad_none_names"]:
        assert k not in G


@pytest.mark.parametrize("name", ["tiny_h64", "small_h128"])
def test_fp32_oracle_close_to_reference(name):
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    labels, masked, lengths, idx = _batch(g)
    loss, pred, _ = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx, dtype=np.float32)
    assert np.abs(pred[_valid(g)] - g["logits"][_valid(g)]).max() < 1e-5
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < 1e-6


@pytest.mark.parametrize("name", ["tiny_h64", "small_h128"])
def test_adamw_trajectory(name):
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    labels, masked, lengths, idx = _batch(g)
    P = {k: v.astype(np.float64) for k, v in sd.items()}
    opt = onp.AdamW(lr=1e-3)
    losses = [onp.train_step(ocfg, P, opt, masked, labels, lengths, idx, dtype=np.float64) for _ in g["losses"]]
    assert np.allclose(losses, g["losses"], rtol=2e-6, atol=0)
    for k in g["param_names"]:
        # 3 steps at lr 1e-3 move a weight by up to 3e-3; Adam's m/sqrt(v) amplifies fp32 grad noise of
        # near-zero gradients, so allow 1% of the largest possible move
        assert np.abs(P[k] - g["final/" + k]).max() < 3e-5, k
    # the pooler never moves (no grad -> AdamW skips it, weight decay included)
    assert np.array_equal(P["encoder.pooler.weight"].astype(np.float32), sd["encoder.pooler.weight"])


@pytest.mark.parametrize("name", ["tiny_h64_multitask", "small_h128_multitask"])
def test_multitask_heads(name):
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    am = onp.attention_mask_from_lengths(g["lengths"])
    (ph, tok), h, _ = onp.model_forward(ocfg, sd, g["masked"], am, dtype=np.float64)
    v = _valid(g)
    assert np.abs(ph[v] - g["logits"][v]).max() < 1e-5
    assert np.abs(tok[v] - g["token_logits"][v]).max() < 1e-5
    assert np.abs(h[v] - g["hidden"][v]).max() < 1e-5


@pytest.mark.parametrize("name,steps", [("real_s128_b8", 2), ("real_s512_b2_ragged", 1)])
def test_real_model_probes(name, steps):
    """768/12 config of configs/config.yml: probe logits, loss, grad norms, loss trajectory."""
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    labels, masked, lengths, idx = _batch(g)
    loss, pred, G = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx, dtype=np.float32)
    pb, ps = g["probe_b"], g["probe_s"]
    assert np.abs(pred[pb, ps] - g["probe_logits"]).max() < 1e-4
    v = _valid(g)
    assert np.abs(pred.sum(-1)[v] - g["logit_row_sums"][v]).max() < 2e-3
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < 1e-5
    for k, ref in zip(g["grad_names"], g["grad_l2"]):
        got = np.sqrt((G[k].astype(np.float64) ** 2).sum())
        assert abs(got - ref) <= 1e-3 * ref + 1e-7, k
        flat = G[k].reshape(-1)
        assert np.abs(flat[g["gprobe_idx/" + k]] - g["gprobe_val/" + k]).max() <= 1e-3 * np.abs(G[k]).max() + 1e-8, k
    if steps > 1:
        P = {k: v.copy() for k, v in sd.items()}
        opt = onp.AdamW(lr=7e-5)
        losses = [float(onp.train_step(ocfg, P, opt, masked, labels, lengths, idx)) for _ in range(steps)]
        assert np.allclose(losses, g["losses"][:steps], rtol=1e-5)


@pytest.mark.parametrize("name", ["tiny_h64_dualloss", "small_h128_dualloss"])
def test_dual_head_loss_grads_trajectory(name):
    """Dual-head training (phoneme + token loss) on the reference's MultiTaskModel under torch autograd:
    the token-loss formula is upstream PL-BERT's (the reference has none, see oracle.albert_np.token_loss)."""
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    labels, masked, lengths, idx = _batch(g)
    tok = g["token_ids"]
    loss, (ph, tk), G = onp.loss_and_grads(ocfg, sd, masked, labels, lengths, idx, dtype=np.float64, token_ids=tok)
    assert abs(loss - g["losses"][0]) / g["losses"][0] < 1e-6
    pl, _ = onp.phoneme_loss(ph, np.asarray(labels), lengths, idx)
    tl, _ = onp.token_loss(tk, tok, lengths)
    assert np.allclose([pl, tl], g["loss_parts"][0], rtol=1e-6)
    for k in g["grad_names"]:
        ref = g["grad/" + k]
        assert np.abs(G[k] - ref).max() < 2e-6 + 1e-4 * np.abs(ref).max(), k
    for k in g["grad_none_names"]:
        assert k not in G
    P = {k: v.astype(np.float64) for k, v in sd.items()}
    opt = onp.AdamW(lr=1e-3)
    losses = [onp.train_step(ocfg, P, opt, masked, labels, lengths, idx, dtype=np.float64, token_ids=tok)
              for _ in g["losses"]]
    assert np.allclose(losses, g["losses"], rtol=2e-6, atol=0)
    for k in g["param_names"]:
        assert np.abs(P[k] - g["final/" + k]).max() < 3e-5, k
