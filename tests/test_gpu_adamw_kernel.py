"""AdamW pinned at kernel level (-m gpu): the SAME fp32 gradients go into plb_adamw_step and into
torch.optim.AdamW (train.py:272: lr, default betas / eps / weight_decay 0.01 on every parameter); parameters and
both moments must agree to 1e-6 relative after 5 steps, including the 1/world gradient scale of a
data-parallel step. A missing decay term (lr*wd*p), a wrong eps placement or bias correction fails this."""
import numpy as np
import pytest
import torch

from gpu_util import rel_l2
import plbert_amd
from plbert_amd.engine import HipEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("grad_scale,lr,wd", [(1.0, 7e-5, 0.01), (0.125, 1e-3, 0.01), (1.0, 1e-3, 0.0)])
def test_adamw_matches_torch(grad_scale, lr, wd):
    cfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                  intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    eng = HipEngine(cfg, 188, 0, max_batch=2, max_seq=32)
    eng.load_state_dict(plbert_amd.deterministic_state_dict(cfg, 188, seed=3))
    eng._bind()
    n = eng.trainable
    p0 = eng.params[:n].clone()
    p_ref = torch.nn.Parameter(eng.params[:n].clone())
    opt = torch.optim.AdamW([p_ref], lr=lr, weight_decay=wd)
    gen = torch.Generator(device="cuda").manual_seed(5)
    pool_before = eng.params[n:].clone()
    for step in range(1, 6):
        # gradients spanning many magnitudes (sqrt(v) + eps is where an eps bug hides), some exactly zero
        g = torch.randn(n, device="cuda", generator=gen) * torch.exp(torch.randn(n, device="cuda", generator=gen) * 3 - 6)
        g[::97] = 0.0
        eng.grads[:n].copy_(g)
        p_ref.grad = g * grad_scale
        opt.step()
        eng.adamw_step(step, lr=lr, weight_decay=wd, grad_scale=grad_scale)
    torch.cuda.synchronize()
    st = opt.state[p_ref]
    assert rel_l2(eng.params[:n], p_ref.detach()) < 1e-6
    assert float((eng.params[:n] - p_ref.detach()).abs().max()) < 1e-6 * float(p_ref.detach().abs().max()) + 1e-9
    assert rel_l2(eng.exp_avg[:n], st["exp_avg"]) < 1e-6
    assert rel_l2(eng.exp_avg_sq[:n], st["exp_avg_sq"]) < 1e-6
    # the update moved the parameters by about lr per step: the comparison above is not vacuous
    moved = (eng.params[:n] - p0).abs()
    assert float(moved.max()) > 2 * lr and rel_l2(eng.params[:n] - p0, p_ref.detach() - p0) < 1e-4
    # nothing outside the trainable range moved (the pooler never trains: train.py:383-390 never reads it)
    assert torch.equal(eng.params[n:], pool_before)
    # the bf16 compute copy follows the fp32 master weights: a forward after the step sees the new parameters
    ids = np.random.RandomState(0).randint(1, 180, size=(2, 32))
    _, ph, _ = eng.forward(ids)
    eng2 = HipEngine(cfg, 188, 0, max_batch=2, max_seq=32)
    eng2.load_state_dict(eng.state_dict())
    _, ph2, _ = eng2.forward(ids)
    torch.cuda.synchronize()
    assert torch.equal(ph, ph2)
