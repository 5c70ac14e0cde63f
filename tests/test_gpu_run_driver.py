"""Run management around the step (-m gpu; SURVEY.md §8(f) N1: train.py:133-210, 261-379, 412-425): a run directory is
created with a copy of the config, validation runs first, ``step_N.pth`` files in the reference's format appear every
``save_interval``, and calling ``train`` again on the same run name RESUMES from the latest of them."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import load_golden
import plbert_amd
from plbert_amd import data as pdata
from plbert_amd import run as prun

pytestmark = pytest.mark.gpu


def _config(tmp_path, num_steps, **training):
    cfg = {"training_params": dict(output_dir=str(tmp_path / "runs"), batch_size=4, mixed_precision="fp16", learning_rate=1e-3,
                                   num_steps=num_steps, save_interval=3, log_interval=2, training_dataset="unused",
                                   split="train", **training),
           "dataset_params": dict(max_seq_length=64, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1,
                                  word_separator=87),
           "model_params": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, max_position_embeddings=512,
                                num_hidden_layers=2, embedding_size=64, pretrained_model="", dropout=0.1)}
    path = tmp_path / "config.yml"
    path.write_text(yaml.safe_dump(cfg))
    return str(path)


def _docs():
    g = load_golden("masking")
    return [{"phonemes": d.split("\x1f")} for d in g["docs"] if len(d) > 0] * 12


@pytest.mark.parametrize("device_masking", [False, True])
def test_run_directory_checkpoints_and_resume(tmp_path, device_masking):
    path = _config(tmp_path, 6, device_masking=device_masking)
    args = {"config_path": path, "run_name": "r"}
    torch.manual_seed(0)
    pdata.seed_reference_streams(1)
    trainer, step, epoch = prun.train(args, dataset=_docs())
    run_dir = tmp_path / "runs" / "r"
    assert step == 6 and (run_dir / "config.yml").exists()
    assert sorted(f for f in os.listdir(run_dir) if f.startswith("step_")) == ["step_3.pth", "step_6.pth"]
    ck = torch.load(run_dir / "step_6.pth", map_location="cpu", weights_only=False)
    assert set(ck) == {"net", "step", "epoch", "optimizer"} and ck["step"] == 6
    assert "encoder.embeddings.word_embeddings.weight" in ck["net"] and "phoneme_predictor.weight" in ck["net"]
    # torch's own AdamW over parameters in the REFERENCE's model.parameters() order accepts the optimizer entry, and
    # every moment has the shape of the parameter at its index (tests/test_checkpoint_order.py pins the order itself)
    from plbert_amd.checkpoint import reference_param_names
    order = reference_param_names({k: None for k in ck["net"]})
    ps = [torch.nn.Parameter(ck["net"][n].clone().float()) for n in order]
    topt = torch.optim.AdamW(ps, lr=1e-3)
    topt.load_state_dict(ck["optimizer"])
    for p_ in ps:
        if p_ in topt.state:
            assert topt.state[p_]["exp_avg"].shape == p_.shape
    recs = [json.loads(l) for l in (run_dir / "metrics.jsonl").read_text().splitlines()]
    assert "val_phoneme_loss" in recs[0] and recs[0]["step"] == 0                      # validation before the first step
    assert sum("val_phoneme_loss" in r for r in recs) == 3                             # step 0, 3, 6
    assert [r["step"] for r in recs if "phoneme_loss" in r] == [1, 2, 3, 4, 5, 6]
    assert all("phoneme_loss_avg" in r for r in recs if "phoneme_loss" in r and r["step"] >= 2)
    w6 = trainer.engine.state_dict()["phoneme_predictor.weight"].cpu()
    assert torch.equal(w6, ck["net"]["phoneme_predictor.weight"])
    # same run name, larger budget: resumes at step 6 with the saved weights and optimizer state
    cfg = yaml.safe_load(open(run_dir / "config.yml"))
    cfg["training_params"]["num_steps"] = 8
    (run_dir / "config.yml").write_text(yaml.safe_dump(cfg))
    trainer2, step2, _ = prun.train(args, dataset=_docs())
    assert step2 == 8 and trainer2.step_count == 8
    recs2 = [json.loads(l) for l in (run_dir / "metrics.jsonl").read_text().splitlines()][len(recs):]
    assert recs2[0]["step"] == 6 and "val_phoneme_loss" in recs2[0]                    # validation at the resumed step
    assert [r["step"] for r in recs2 if "phoneme_loss" in r] == [7, 8]
    # a folder without a config copy is cleaned and starts fresh
    os.remove(run_dir / "config.yml")
    _, _, resuming = prun.setup_config_and_directories(args, path)
    assert not resuming and not [f for f in os.listdir(run_dir) if f.startswith("step_")]


@pytest.mark.parametrize("num_workers", [0, 2])
def test_deferred_loss_readback_logs_what_the_immediate_one_logs(tmp_path, monkeypatch, num_workers):
    """train_loop reads every step's loss one step late (run._LossReader: no stall between steps); the records — losses,
    running means, validation losses, their order — and the final weights must be those of the loop that reads each loss
    before enqueuing the next step, as the reference does (train.py:381-410)."""
    out = {}
    loop = prun.train_loop
    for deferred in (True, False):
        sub = tmp_path / ("deferred" if deferred else "immediate")
        sub.mkdir()
        path = _config(sub, 7, device_masking=True, num_workers=num_workers)
        monkeypatch.setattr(prun, "train_loop", lambda *a, _d=deferred, **kw: loop(*a, deferred_readback=_d, **kw))
        torch.manual_seed(0)
        pdata.seed_reference_streams(1)
        trainer, step, _ = prun.train({"config_path": path, "run_name": "r"}, dataset=_docs())
        assert step == 7
        recs = [json.loads(l) for l in (sub / "runs" / "r" / "metrics.jsonl").read_text().splitlines()]
        out[deferred] = (recs, trainer.engine.params.clone())
        del trainer
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1])
    assert [r["step"] for r in out[True][0] if "phoneme_loss" in r] == list(range(1, 8))


def _launched_rank(rank, world, port, args, docs, out, env=None):
    # what torchrun / accelerate launch export: train() must join the group and pick its GPU from these alone
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), **(env or {}))
    import torch.distributed as dist
    from plbert_amd import data as pdata_, run as prun_

    torch.manual_seed(0)
    pdata_.seed_reference_streams(1)
    trainer, step, epoch = prun_.train(args, dataset=docs)
    out[rank] = (step, trainer.world, trainer.comm, str(trainer.engine.device),
                 trainer.engine.state_dict()["phoneme_predictor.weight"].cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded,exchange", [(False, "torch"), (True, "torch"), (False, "rccl")])
def test_train_under_a_launcher_two_ranks(tmp_path, fake_lib, sharded, exchange):
    """``train(args)`` started as two ranks (RANK / WORLD_SIZE / LOCAL_RANK in the environment, nothing else): each rank
    joins the process group itself and takes GPU LOCAL_RANK (both map to the box's one GPU here), rank 0 alone creates
    the run directory, writes metrics and checkpoints, the ragged last validation batch (7 documents, batch 4, two
    ranks) is shared out instead of raising, and the replicas end bit-identical. Every rank's batches come through the
    DeviceFeeder (copy stream, double-buffered slots). ``sharded``: training_params.shard_samples — each rank loads and masks
    only its own half of the samples (batches of 2), with the masking applied on the device. ``exchange == "rccl"``: the
    ranks exchange through the ENGINE's communicator (what a run with one GPU per rank uses: pieces inside the backward,
    health word, deferred loss read-back) over the stand-in library of tests/fake_rccl.cpp, PLBERT_COMM=rccl."""
    import torch.multiprocessing as mp

    path = _config(tmp_path, 4, **(dict(shard_samples=True, device_masking=True, num_workers=2) if sharded else {}))
    args = {"config_path": path, "run_name": "two"}
    docs = (_docs() * 3)[:140]                            # 5 % validation split = 7 documents: batches of 4 and 3
    assert len(docs) == 140
    port = 29700 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        env = dict(PLBERT_COMM="rccl", PLBERT_RCCL_LIB=fake_lib, FAKE_RCCL_TIMEOUT_S="120") if exchange == "rccl" else None
        mp.spawn(_launched_rank, args=(2, port, args, docs, out, env), nprocs=2, join=True)
        res = dict(out)
    assert res[0][0] == res[1][0] == 4 and res[0][1] == res[1][1] == 2
    assert res[0][2] == res[1][2] == exchange             # two ranks on one device: the gloo fallback unless told otherwise
    assert np.array_equal(res[0][4], res[1][4])
    run_dir = tmp_path / "runs" / "two"
    assert sorted(f for f in os.listdir(run_dir) if f.startswith("step_")) == ["step_3.pth"]
    recs = [json.loads(l) for l in (run_dir / "metrics.jsonl").read_text().splitlines()]
    assert [r["step"] for r in recs if "phoneme_loss" in r] == [1, 2, 3, 4]    # one writer
    assert sum("val_phoneme_loss" in r for r in recs) == 2
