import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True)


def golden_cfg(g):
    """(oracle Config, product AlbertConfig kwargs, state dict) from a model fixture."""
    from oracle.albert_np import Config
    import plbert_amd

    kw = {k: int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    pcfg = plbert_amd.AlbertConfig(**kw)
    nph, ntok = int(g["num_phonemes"]), int(g["num_tokens"])
    # "init" names the generator the capture used (oracle/gen_golden.py: capture_model); fixtures older than it: deterministic
    init = str(g["init"]) if "init" in g.files else "deterministic"
    gen = plbert_amd.reference_init_state_dict if init == "reference" else plbert_amd.deterministic_state_dict
    sd = gen(pcfg, nph, ntok, seed=int(g["seed"]))
    ocfg = Config(vocab_size=pcfg.vocab_size, embedding_size=pcfg.embedding_size, hidden_size=pcfg.hidden_size,
                  num_attention_heads=pcfg.num_attention_heads, intermediate_size=pcfg.intermediate_size,
                  num_hidden_layers=pcfg.num_hidden_layers, num_phonemes=nph, num_tokens=ntok)
    return ocfg, pcfg, sd


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def fake_lib(tmp_path_factory):
    """tests/fake_rccl.cpp built once per session: RCCL's entry points over POSIX shared memory between processes that share
    the GPU (PLBERT_RCCL_LIB) — the engine's own gradient exchange at world > 1 on a one-GPU box."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out,
                        os.path.join(root, "tests", "fake_rccl.cpp"), "-lrt", "-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return out
