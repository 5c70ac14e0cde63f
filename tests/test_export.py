"""Export for the encoder's consumers: the directory of convert_to_hf.py:16-64 and the key
recipe of README.md:57-63, checked with the installed ``transformers`` AlbertModel as the consumer and the oracle /
reference-captured fixtures as the expected values. CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
from oracle import albert_np as onp
from plbert_amd import export, symbols

transformers = pytest.importorskip("transformers")


def _yaml_config(g):
    kw = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    assert kw.pop("vocab_size") == len(symbols)
    kw.update(pretrained_model="", dropout=0.1)  # keys of configs/config.yml the encoder config must ignore
    return {"log_dir": "Checkpoint", "batch_size": 2, "model_params": kw,
            "preprocess_params": {"tokenizer": "aubmindlab/bert-base-arabertv2"}}


def _checkpoint(g, tmp_path, ddp_prefix=True):
    _, _, sd = golden_cfg(g)
    pre = "module." if ddp_prefix else ""
    net = {pre + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    path = os.path.join(tmp_path, "step_7.pth")
    torch.save({"net": net, "step": 7, "epoch": 1, "optimizer": None}, path)
    return path, sd


def _mask(g):
    S = g["labels"].shape[1]
    return (np.arange(S)[None, :] < g["lengths"][:, None]).astype(np.int64)


@pytest.mark.parametrize("name", ["tiny_h64", "tiny_h64_multitask"])
def test_exported_directory_loads_into_transformers_and_matches_the_reference(name, tmp_path):
    g = load_golden(name)
    path, sd = _checkpoint(g, tmp_path)
    out = os.path.join(tmp_path, "hf")
    paths = export.export_pretrained(path, _yaml_config(g), out)
    assert sorted(os.listdir(out)) == ["config.json", "config.yml", "model.safetensors", "pl_bert_full_model.pt",
                                       "training_metadata.txt"]
    meta = open(paths["metadata"]).read()
    assert f"Original checkpoint: {path}" in meta and "Step: 7" in meta and "Epoch: 1" in meta
    cj = json.load(open(paths["config"]))
    assert cj["model_type"] == "albert" and cj["hidden_act"] == "gelu_new" and cj["vocab_size"] == len(symbols)
    assert "pretrained_model" not in cj and "dropout" not in cj

    bert = transformers.AlbertModel.from_pretrained(out, attn_implementation="eager")
    bert.eval()
    own = dict(bert.state_dict())
    for k, v in sd.items():
        if k.startswith("encoder."):
            assert torch.equal(own[k[len("encoder."):]], torch.from_numpy(v)), k
    ids, am = torch.from_numpy(g["masked"]), torch.from_numpy(_mask(g))
    with torch.no_grad():
        hidden = bert(ids, attention_mask=am).last_hidden_state.numpy()
    valid = _mask(g).astype(bool)
    ocfg, _, _ = golden_cfg(g)
    want, _ = onp.encoder_forward(ocfg, sd, g["masked"], attention_mask=_mask(g), keep=False)
    assert np.abs(hidden[valid] - want[valid]).max() < 1e-5
    if "hidden" in g.files:  # captured from the reference model itself
        assert np.abs(hidden[valid] - g["hidden"][valid]).max() < 1e-5

    config, full, num_tokens = export.read_exported(out)
    assert num_tokens == int(g["num_tokens"])
    assert set(full) == set(sd) and all(torch.equal(full[k], torch.from_numpy(sd[k])) for k in sd)
    assert config["model_params"]["hidden_size"] == 64


def test_readme_recipe_keys_load_strictly(tmp_path):
    g = load_golden("tiny_h64")
    path, sd = _checkpoint(g, tmp_path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    enc = export.encoder_state_dict(ck["net"])
    assert "embeddings.word_embeddings.weight" in enc and "pooler.weight" in enc
    assert not any(k.startswith("phoneme_predictor") for k in enc)
    mp = _yaml_config(g)["model_params"]
    bert = transformers.AlbertModel(transformers.AlbertConfig(vocab_size=len(symbols), **mp))
    res = bert.load_state_dict(enc, strict=True)  # README.md:63
    assert not res.missing_keys and not res.unexpected_keys


def test_export_refuses_what_it_cannot_describe(tmp_path):
    g = load_golden("tiny_h64")
    _, _, sd = golden_cfg(g)
    heads_only = {k: torch.from_numpy(v) for k, v in sd.items() if not k.startswith("encoder.")}
    with pytest.raises(ValueError, match="encoder"):
        export.export_pretrained(heads_only, _yaml_config(g), os.path.join(tmp_path, "a"))
    cfg = _yaml_config(g)
    cfg["model_params"]["hidden_size"] = 128
    with pytest.raises(ValueError, match="shape"):
        export.export_pretrained({k: torch.from_numpy(v) for k, v in sd.items()}, cfg, os.path.join(tmp_path, "b"))


def test_command_line(tmp_path):
    import yaml
    g = load_golden("tiny_h64")
    path, _ = _checkpoint(g, tmp_path, ddp_prefix=False)
    cfgp = os.path.join(tmp_path, "config.yml")
    with open(cfgp, "w") as f:
        yaml.dump(_yaml_config(g), f)
    out = os.path.join(tmp_path, "cli")
    export.main(["--checkpoint_path", path, "--config_path", cfgp, "--output_dir", out])
    assert os.path.isfile(os.path.join(out, "model.safetensors"))
