"""Dual-head training (-m gpu): phoneme loss + token (grapheme) loss through plb_loss_fwd_bwd_dual, against
(a) the goldens captured from torch autograd over the reference's MultiTaskModel and (b) the oracle on
seeded shapes, incl. a vocabulary that is not a multiple of the 256-column GEMM tile and the 64 k case."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
from gpu_util import rel_l2
from oracle import albert_np as onp
import plbert_amd
from plbert_amd.engine import HipEngine
from plbert_amd.train import PLBertTrainer

pytestmark = pytest.mark.gpu

KEY_BIAS = "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.key.bias"


def _inputs(g):
    idx = [list(map(int, x)) for x in g["index"]]
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    return g["masked"], g["labels"], g["lengths"].astype(np.int32), off, flat, int(off[-1]), g["token_ids"]


def test_dual_loss_and_grads_golden():
    g = load_golden("small_h128_dualloss")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    eng = HipEngine(pcfg, 188, int(g["num_tokens"]), max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    masked, labels, lens, off, flat, n, tok = _inputs(g)
    loss = eng.loss_fwd_bwd(masked, labels, lens, off, flat, n, token_ids=tok)
    torch.cuda.synchronize()
    ref = float(g["losses"][0])
    assert abs(float(loss.item()) - ref) / ref < 1e-3                      # north_star: loss within 1e-3 relative
    assert np.allclose(eng.loss_parts.cpu().numpy(), g["loss_parts"][0], rtol=1e-3)
    for k in g["grad_names"]:
        k = str(k)
        if k == KEY_BIAS:  # true gradient is 0 (softmax shift invariance); covered in test_gpu_engine.py
            continue
        got = eng.view(k, of=eng.grads).cpu()
        assert rel_l2(got, torch.from_numpy(g["grad/" + k])) < 4e-2, k   # bf16 backward, per-tensor relative L2


def test_dual_adamw_trajectory_golden():
    g = load_golden("small_h128_dualloss")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    eng = HipEngine(pcfg, 188, int(g["num_tokens"]), max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    masked, labels, lens, off, flat, n, tok = _inputs(g)
    losses = []
    for step in range(1, len(g["losses"]) + 1):
        losses.append(float(eng.loss_fwd_bwd(masked, labels, lens, off, flat, n, token_ids=tok).item()))
        eng.adamw_step(step, lr=1e-3)
    torch.cuda.synchronize()
    assert np.allclose(losses, g["losses"], rtol=2e-3)
    for k in ("token_predictor.weight", "token_predictor.bias", "phoneme_predictor.weight",
              "encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn.weight", "encoder.embeddings.word_embeddings.weight"):
        d_got = eng.view(k).cpu() - torch.from_numpy(sd[k])
        d_ref = torch.from_numpy(g["final/" + k] - sd[k])
        assert rel_l2(d_got, d_ref) < 0.12, (k, rel_l2(d_got, d_ref))   # bf16 gradient noise through Adam (measured <= 0.1)
    # a phoneme-only step afterwards leaves the token head alone (no gradient -> no update, as torch)
    before = eng.view("token_predictor.weight").clone()
    eng.loss_fwd_bwd(masked, labels, lens, off, flat, n)
    eng.adamw_step(len(g["losses"]) + 1, lr=1e-3)
    torch.cuda.synchronize()
    assert torch.equal(eng.view("token_predictor.weight"), before)


def test_dual_full_size_against_reference_probes():
    """BASELINE configs[1] names a dual-head loss at 32 x 512: the reference's MultiTaskModel (768 / 12, a 5,000-token head
    — not a multiple of the 256-column tile of the fused GEMM + cross-entropy passes) on bench.py's own batch and
    initialisation, gradients by torch autograd over the reference model, AdamW lr 7e-5 (tests/golden/
    real_s512_b32_dualloss.npz, oracle/gen_golden.py fullsize_dual; probes only). Both heads' logits on 16 probe rows, the
    loss and its two parts, every gradient tensor's norm and 8 of its elements, the 2-step trajectory."""
    g = load_golden("real_s512_b32_dualloss")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    eng = HipEngine(pcfg, 188, int(g["num_tokens"]), max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    masked, labels, lens, off, flat, n, tok = _inputs(g)
    _, ph, tk = eng.forward(masked, lens, want_token=True)
    pb, ps = torch.as_tensor(g["probe_b"]), torch.as_tensor(g["probe_s"])
    assert np.abs(ph[pb, ps].cpu().numpy() - g["probe_logits"]).max() < 3e-2
    assert np.abs(tk[pb, ps].cpu().numpy() - g["probe_token_logits"]).max() < 3e-2
    del tk
    losses = []
    for step in range(1, len(g["losses"]) + 1):
        loss = eng.loss_fwd_bwd(masked, labels, lens, off, flat, n, token_ids=tok)
        if step == 1:
            torch.cuda.synchronize()
            assert np.allclose(eng.loss_parts.cpu().numpy(), g["loss_parts"][0], rtol=1e-3)
            for k, ref in zip(g["grad_names"], g["grad_l2"]):
                k = str(k)
                if k == KEY_BIAS:
                    continue
                got = eng.view(k, of=eng.grads)
                assert abs(float(got.double().norm()) - ref) <= 3e-2 * ref + 1e-6, (k, float(got.double().norm()), ref)
                flatg = got.flatten().cpu().numpy()
                assert np.abs(flatg[g["gprobe_idx/" + k]] - g["gprobe_val/" + k]).max() <= 0.1 * np.abs(flatg).max() + 1e-7, k
        losses.append(float(loss.item()))
        eng.adamw_step(step, lr=float(g["lr"]))
    assert np.allclose(losses, g["losses"], rtol=1e-3), (losses, g["losses"])
    assert eng.status()["ln_exchange_timeouts"] == 0


@pytest.mark.parametrize("num_tokens,B,S,lengths", [(1000, 3, 64, [64, 40, 9]), (2304, 2, 128, [128, 128])])
def test_dual_against_oracle(num_tokens, B, S, lengths):
    """Seeded shapes vs the fp32 oracle: vocabulary not a multiple of 256 (padded columns), ragged lengths,
    and a sample without masked phonemes."""
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                   intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, num_tokens, seed=5)
    labels, masked, lens, idx = plbert_amd.synthetic_batch(B, S, seed=77)
    lens = list(lengths)
    idx = [[i for i in ix if i < L] for ix, L in zip(idx, lens)]
    idx[-1] = []                                              # one sample contributes no phoneme loss
    rs = np.random.RandomState(3)
    tok = rs.randint(0, num_tokens, size=(B, S)).astype(np.int64)
    ocfg = onp.Config(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                      num_hidden_layers=2, max_position_embeddings=512, num_phonemes=188, num_tokens=num_tokens)
    loss_ref, _, G = onp.loss_and_grads(ocfg, sd, masked, labels, lens, idx, token_ids=tok)
    eng = HipEngine(pcfg, 188, num_tokens, max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    off, flat = plbert_amd.masked_indices_to_csr(idx)
    loss = eng.loss_fwd_bwd(masked, labels, np.asarray(lens, np.int32), off, flat, int(off[-1]), token_ids=tok)
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(loss_ref)) / float(loss_ref) < 1e-3
    for k in ("token_predictor.weight", "token_predictor.bias", "phoneme_predictor.weight",
              "encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn.weight",
              "encoder.embeddings.word_embeddings.weight"):
        assert rel_l2(eng.view(k, of=eng.grads).cpu(), torch.from_numpy(G[k])) < 4e-2, k


def test_dual_no_masked_phonemes_at_all():
    """n_masked == 0 with token targets: the phoneme term is 0, the token term still trains."""
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2,
                                   intermediate_size=256, num_hidden_layers=2, max_position_embeddings=512)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, 512, seed=6)
    labels, masked, lens, _ = plbert_amd.synthetic_batch(2, 32, seed=78)
    tok = np.random.RandomState(4).randint(0, 512, size=(2, 32)).astype(np.int64)
    ocfg = onp.Config(vocab_size=188, embedding_size=64, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                      num_hidden_layers=2, max_position_embeddings=512, num_phonemes=188, num_tokens=512)
    loss_ref, _, G = onp.loss_and_grads(ocfg, sd, labels, labels, lens, [[], []], token_ids=tok)
    eng = HipEngine(pcfg, 188, 512, max_batch=2, max_seq=32)
    eng.load_state_dict(sd)
    off, flat = plbert_amd.masked_indices_to_csr([[], []])
    loss = eng.loss_fwd_bwd(labels, labels, None, off, flat, 0, token_ids=tok)
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(loss_ref)) / float(loss_ref) < 1e-3
    assert float(eng.loss_parts[0].item()) == 0.0
    assert float(eng.view("phoneme_predictor.weight", of=eng.grads).abs().max()) == 0.0
    assert rel_l2(eng.view("token_predictor.weight", of=eng.grads).cpu(), torch.from_numpy(G["token_predictor.weight"])) < 4e-2


def test_dual_64k_vocabulary_properties():
    """The believed production vocabulary (64 000 tokens) at the real hidden size: the token loss
    equals torch's cross-entropy of the engine's own fp32 token logits (1e-3 relative); the token-bias gradient
    sums to 0 (softmax rows sum to 1); the token-weight gradient matches a torch evaluation on probe classes."""
    NT = 64000
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                   max_position_embeddings=512, num_hidden_layers=2)
    B, S = 4, 256
    tr = PLBertTrainer(pcfg, 188, max_batch=B, max_seq=S, lr=1e-4, seed=0, num_tokens=NT)
    labels, masked, lens, idx = plbert_amd.synthetic_batch(B, S, seed=9)
    tok = np.random.RandomState(5).randint(0, NT, size=(B, S)).astype(np.int64)
    batch = tr.stage_batch(labels, masked, lens, idx, token_ids=tok)
    eng = tr.engine
    loss = tr.loss_and_grads(batch)
    torch.cuda.synchronize()
    parts = eng.loss_parts.cpu().numpy()
    assert abs(float(loss.item()) - parts.sum()) < 1e-4
    gb = eng.view("token_predictor.bias", of=eng.grads).double()
    assert abs(float(gb.sum())) < 2e-3 and float(gb.abs().sum()) > 0.5     # sum_c (p_c - onehot_c) = 0 per row
    # probe: dW[c] for the target classes of the first sample, from fp32 logits recomputed with torch
    hid, _, tk = eng.forward(masked, None, want_hidden=True, want_phoneme=False, want_token=True)
    h = hid.reshape(B * S, -1).double()
    tgt = torch.from_numpy(tok.reshape(-1)).to(h.device)
    ce = float(torch.nn.functional.cross_entropy(tk.reshape(B * S, NT).double(), tgt).item())   # all lengths = S
    assert abs(parts[1] - ce) / ce < 1e-3, (parts, ce)
    assert abs(ce - np.log(NT)) < 0.5                                        # a fresh head is near-uniform
    p = torch.softmax(tk.reshape(B * S, NT).double(), -1)
    rows = torch.arange(B * S, device=h.device)
    p[rows, tgt] -= 1.0
    p /= (B * S)                                                             # w = 1 / (B * len), len = S
    probe = torch.from_numpy(np.unique(tok.reshape(-1))[:64]).to(h.device)
    want = p[:, probe].T @ h
    got = eng.view("token_predictor.weight", of=eng.grads)[probe].double()
    assert rel_l2(got.cpu().float(), want.cpu().float()) < 4e-2
    l0 = float(loss.item())
    for _ in range(3):
        l1 = float(tr.step(batch).item())
    assert l1 < l0                                                           # both heads learn


def test_dual_head_optimizer_state_round_trips_and_keeps_its_own_step():
    """AdamW.state_dict()/load_state_dict() must carry the token head's moments and step. A model that
    trains phoneme-only first and dual-head later has DIFFERENT step counts for the two ranges (torch keeps one per
    parameter); save -> load into a fresh model -> the next dual step is bit-identical to the uninterrupted run."""
    from plbert_amd.train import AdamW, process_batch
    g = load_golden("small_h128_dualloss")
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    NT = int(g["num_tokens"])
    idx = [list(map(int, x)) for x in g["index"]]
    lens = [int(x) for x in g["lengths"]]
    b3 = (torch.from_numpy(g["labels"]), torch.from_numpy(g["masked"]), lens, idx)
    b4 = (torch.from_numpy(g["token_ids"]),) + b3

    def make():
        m = plbert_amd.MultiTaskModel(plbert_amd.AlbertModel(pcfg, max_batch=B, max_seq=S), 188, NT, pcfg.hidden_size)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m, AdamW(m.parameters(), lr=1e-3, model=m)

    def step(m, opt, batch):
        loss = process_batch(m, batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return float(loss.detach())

    m, opt = make()
    step(m, opt, b3); step(m, opt, b3)            # two phoneme-only steps: the token head has no state yet
    st = opt.state_dict()
    names = [n for n, _ in m.named_parameters()]
    tok_idx = [i for i, n in enumerate(names) if n.startswith("token_predictor")]
    assert all(i not in st["state"] for i in tok_idx)
    step(m, opt, b4)                              # first dual step: token head step 1, encoder step 3
    st = opt.state_dict()
    assert all(float(st["state"][i]["step"]) == 1.0 for i in tok_idx)
    assert float(st["state"][names.index("phoneme_predictor.weight")]["step"]) == 3.0
    # torch's own AdamW accepts the saved state (train.py:417-421 'optimizer' entry)
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in m.parameters()]
    torch.optim.AdamW(ref_params, lr=1e-3).load_state_dict(st)
    net = {k: v.detach().clone() for k, v in m.state_dict().items()}
    l_next = step(m, opt, b4)                     # uninterrupted run: step 4 / 2
    # resume in a fresh model
    m2, opt2 = make()
    m2.load_state_dict(net, strict=False)
    opt2.load_state_dict(st)
    assert opt2.step_count == 3 and m2.engine.token_head_steps == 1
    l_res = step(m2, opt2, b4)
    torch.cuda.synchronize()
    assert l_res == l_next
    for (n, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n
