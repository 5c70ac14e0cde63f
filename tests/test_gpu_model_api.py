"""Drop-in boundary (-m gpu): the reference-shaped classes and the reference loop body
(train.py:350-357) running on the HIP engine, checked against goldens captured from the reference."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
import plbert_amd

pytestmark = pytest.mark.gpu


def _make(g, multitask=False):
    ocfg, pcfg, sd = golden_cfg(g)
    B, S = g["labels"].shape
    enc = plbert_amd.AlbertModel(pcfg, max_batch=B, max_seq=S)
    if multitask:
        m = plbert_amd.MultiTaskModel(enc, num_phonemes=int(g["num_phonemes"]), num_tokens=int(g["num_tokens"]),
                                      hidden_size=pcfg.hidden_size)
    else:
        m = plbert_amd.PhonemeOnlyModel(enc, num_phonemes=int(g["num_phonemes"]), hidden_size=pcfg.hidden_size)
    # checkpoints written by DDP carry 'module.' prefixes; the reference strips them (train.py:98)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected
    return m, pcfg, sd


def test_state_dict_names_and_shapes():
    g = load_golden("small_h128_multitask")
    m, pcfg, sd = _make(g, multitask=True)
    got = m.state_dict()
    assert set(got.keys()) == set(sd.keys())
    for k, v in sd.items():
        assert tuple(got[k].shape) == v.shape, k
        assert np.array_equal(got[k].cpu().numpy(), v), k
    assert sum(p.numel() for p in m.parameters()) == sum(v.size for v in sd.values())


def test_forward_signatures_match_reference_outputs():
    g = load_golden("small_h128_multitask")
    m, pcfg, sd = _make(g, multitask=True)
    ids = torch.from_numpy(g["masked"])
    text_mask = plbert_amd.length_to_mask(torch.Tensor(g["lengths"].tolist()))
    am = (~text_mask).int()
    m.eval()
    with torch.no_grad():
        ph, tk = m(ids, attention_mask=am)
        hid = m.encoder(ids, attention_mask=am).last_hidden_state
    assert ph.dtype == torch.float32 and tuple(ph.shape) == g["logits"].shape
    v = (am != 0).numpy()
    assert np.abs(ph.cpu().numpy()[v] - g["logits"][v]).max() < 3e-2
    assert np.abs(tk.cpu().numpy()[v] - g["token_logits"][v]).max() < 3e-2
    assert np.abs(hid.cpu().numpy()[v] - g["hidden"][v]).max() < 6e-2
    with pytest.raises(ValueError):
        bad = am.clone()
        bad[0, 3] = 0  # a hole: not a prefix mask
        m(ids, attention_mask=bad)


def test_reference_loop_body_runs_unchanged():
    """loss = process_batch(...); optimizer.zero_grad(); loss.backward(); optimizer.step()"""
    g = load_golden("small_h128")
    m, pcfg, sd = _make(g)
    opt = plbert_amd.AdamW(m.parameters(), lr=1e-3, model=m)
    batch = (torch.from_numpy(g["labels"]), torch.from_numpy(g["masked"]), [int(x) for x in g["lengths"]],
             [list(map(int, x)) for x in g["index"]])
    m.train()
    losses = []
    for _ in g["losses"]:
        loss = plbert_amd.process_batch(m, batch, criterion=torch.nn.CrossEntropyLoss(), accelerator=None)
        opt.zero_grad()
        loss.backward()
        assert m.phoneme_predictor.weight.grad is not None and m.encoder.pooler.weight.grad is None
        opt.step()
        losses.append(float(loss.item()))
    assert np.allclose(losses, g["losses"], rtol=2e-3)
    # validation path (train.py:288-304): eval + no_grad gives a plain number and changes nothing
    before = m.phoneme_predictor.weight.detach().clone()
    m.eval()
    with torch.no_grad():
        vl = plbert_amd.process_batch(m, batch, None, None)
    assert not vl.requires_grad and torch.equal(before, m.phoneme_predictor.weight.detach())
    # optimizer state round trip ('optimizer' entry of the checkpoint dict, train.py:417-421)
    st = opt.state_dict()
    opt2 = plbert_amd.AdamW(m.parameters(), lr=1e-3, model=m)
    opt2.load_state_dict(st)
    assert opt2.step_count == opt.step_count


def test_checkpoint_files_are_interchangeable_with_torch(tmp_path):
    """step_N.pth layout {'net','step','epoch','optimizer'} (train.py:412-425): written here, the optimizer
    entry loads into torch.optim.AdamW over same-shaped parameters; a file with DDP 'module.' prefixes and a
    torch AdamW state loads back here and training resumes on the golden loss trajectory."""
    g = load_golden("small_h128")
    m, pcfg, sd = _make(g)
    opt = plbert_amd.AdamW(m.parameters(), lr=1e-3, model=m)
    batch = (torch.from_numpy(g["labels"]), torch.from_numpy(g["masked"]), [int(x) for x in g["lengths"]],
             [list(map(int, x)) for x in g["index"]])
    losses = []
    for step in range(2):
        loss = plbert_amd.process_batch(m, batch)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.item()))
    path = plbert_amd.save_checkpoint(m, opt, 2, str(tmp_path), None, current_epoch=1)
    assert plbert_amd.find_latest_checkpoint(str(tmp_path)) == (True, 2)
    ck = torch.load(path, weights_only=False)
    assert set(ck.keys()) == {"net", "step", "epoch", "optimizer"} and ck["step"] == 2 and ck["epoch"] == 1
    # torch's optimizer accepts the saved state for parameters of the same shapes / order
    shadow = [torch.nn.Parameter(p.detach().cpu().clone()) for p in m.parameters()]
    topt = torch.optim.AdamW(shadow, lr=1e-3)
    topt.load_state_dict(ck["optimizer"])
    names = [n for n, _ in m.named_parameters()]
    i_head = names.index("phoneme_predictor.weight")
    assert torch.allclose(topt.state[shadow[i_head]]["exp_avg"], m.engine.view("phoneme_predictor.weight", of=m.engine.exp_avg).cpu())
    assert shadow[names.index("encoder.pooler.weight")] not in topt.state      # never stepped
    # a reference-style file (DDP prefixes, torch optimizer state) resumes here
    ck2 = {"net": {"module." + k: v for k, v in ck["net"].items()}, "step": 2, "epoch": 1, "optimizer": topt.state_dict()}
    torch.save(ck2, str(tmp_path / "step_7.pth"))
    assert plbert_amd.find_latest_checkpoint(str(tmp_path)) == (True, 7)
    m2, _, _ = _make(g)
    opt_b = plbert_amd.AdamW(m2.parameters(), lr=1e-3, model=m2)
    plbert_amd.load_checkpoint(m2, opt_b, str(tmp_path / "step_7.pth"), None)
    assert opt_b.step_count == 2
    loss3 = plbert_amd.process_batch(m2, batch)
    assert abs(float(loss3.item()) - float(g["losses"][2])) / float(g["losses"][2]) < 2e-3


def test_standalone_encoder_loads_stripped_checkpoint():
    """README.md:49-66 consumer: strip 'module.' and 'encoder.' and load into AlbertModel."""
    g = load_golden("small_h128")
    ocfg, pcfg, sd = golden_cfg(g)
    enc = plbert_amd.AlbertModel(pcfg, max_batch=3, max_seq=40)
    stripped = {k[len("encoder."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("encoder.")}
    enc.load_state_dict(stripped, strict=False)
    am = (~plbert_amd.length_to_mask(torch.Tensor(g["lengths"].tolist()))).int()
    out = enc(torch.from_numpy(g["masked"]), attention_mask=am)
    v = (am != 0).numpy()
    assert np.abs(out.last_hidden_state.cpu().numpy()[v] - g["hidden"][v]).max() < 6e-2
    assert tuple(out.pooler_output.shape) == (3, pcfg.hidden_size)


def test_dual_head_loop_body_on_multitask_model():
    """The reference loop body on MultiTaskModel with the 5-tuple Collater batch (dataloader.py:200-223):
    both heads receive gradients, the pooler none; losses follow the reference-captured trajectory."""
    g = load_golden("small_h128_dualloss")
    m, pcfg, sd = _make(g, multitask=True)
    opt = plbert_amd.AdamW(m.parameters(), lr=1e-3, model=m)
    batch = (torch.from_numpy(g["token_ids"]), torch.from_numpy(g["labels"]), torch.from_numpy(g["masked"]),
             [int(x) for x in g["lengths"]], [list(map(int, x)) for x in g["index"]])
    m.train()
    losses = []
    for _ in g["losses"]:
        loss = plbert_amd.process_batch(m, batch, criterion=torch.nn.CrossEntropyLoss(), accelerator=None)
        opt.zero_grad()
        loss.backward()
        assert m.token_predictor.weight.grad is not None and m.phoneme_predictor.weight.grad is not None
        assert m.encoder.pooler.weight.grad is None
        opt.step()
        losses.append(float(loss.item()))
    assert np.allclose(losses, g["losses"], rtol=2e-3)
    st = opt.state_dict()
    names = [n for n, _ in m.named_parameters()]
    assert names.index("token_predictor.weight") in st["state"]       # the token head now has optimizer state
    assert names.index("encoder.pooler.weight") not in st["state"]
    with pytest.raises(ValueError):
        bad = torch.from_numpy(g["token_ids"]).clone()
        bad[0, 0] = int(g["num_tokens"])                               # out-of-range class id
        plbert_amd.process_batch(m, (bad,) + batch[1:], None, None)


@pytest.mark.parametrize("multitask", [False, True])
def test_export_and_load_pl_bert_model_round_trip(multitask, tmp_path):
    """convert_to_hf.py:16-102 on the native path: model → directory → model, logits bit-identical; the
    directory's encoder is what the reference model computed (fixture) within the bf16 bound."""
    from plbert_amd import export
    g = load_golden("small_h128_multitask" if multitask else "small_h128")
    m, pcfg, sd = _make(g, multitask=multitask)
    kw = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    kw.pop("vocab_size")
    config = {"model_params": dict(kw, pretrained_model="", dropout=0.1)}
    export.export_pretrained({"module." + k: v for k, v in m.state_dict().items()}, config, str(tmp_path), step=3, epoch=0)
    B, S = g["labels"].shape
    m2 = export.load_pl_bert_model(str(tmp_path), max_batch=B, max_seq=S)
    assert type(m2).__name__ == ("MultiTaskModel" if multitask else "PhonemeOnlyModel") and not m2.training
    ids = torch.from_numpy(g["masked"])
    am = (~plbert_amd.length_to_mask(torch.Tensor(g["lengths"].tolist()))).int()
    m.eval()
    with torch.no_grad():
        a, b = m(ids, attention_mask=am), m2(ids, attention_mask=am)
    a, b = (a, b) if multitask else ((a,), (b,))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    v = (am != 0).numpy()
    assert np.abs(b[0].cpu().numpy()[v] - g["logits"][v]).max() < 3e-2
