"""The drop-in caller end to end (SURVEY.md §8(f) N1 + N3): ``plbert_amd.run.train`` — the counterpart of the reference's
``train.train(args)`` (train.py:133-172) — on synthetic documents with the reference's own config.yml values (model
768 / 12 / 2048, lr 7e-5, max_seq_length 512, log_interval 10), worker processes drawing the masking decisions, the
device-side masking, a validation pass and a ``step_N.pth`` file every ``save_interval`` steps. Prints the sustained
rate of the whole call (wall clock of ``train`` after the model is built) beside the resident-batch step of bench.py.
   python tools/run_end_to_end.py [batch 32|96] [steps] [workers] [bf16|fp8]"""
import json
import os
import sys
import tempfile
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synthetic_documents)
from plbert_amd import data as pdata  # noqa: E402
from plbert_amd import run as prun  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
fp8 = len(sys.argv) > 4 and sys.argv[4] == "fp8"
save_interval = 500

tmp = tempfile.mkdtemp(prefix="plb_e2e_")
cfg = {"training_params": dict(output_dir=os.path.join(tmp, "runs"), batch_size=B, mixed_precision="fp16", learning_rate=7e-5,
                               num_steps=steps, save_interval=save_interval, log_interval=10, training_dataset="unused",
                               split="train", num_workers=workers, device_masking=True),
       "dataset_params": dict(max_seq_length=512, word_pred_prob=0.15, phoneme_mask_prob=0.8, replace_prob=0.1, word_separator=87),
       "model_params": dict(hidden_size=768, num_attention_heads=12, intermediate_size=2048, max_position_embeddings=512,
                            num_hidden_layers=12, pretrained_model="", dropout=0.1)}
path = os.path.join(tmp, "config.yml")
with open(path, "w") as f:
    yaml.safe_dump(cfg, f)
docs = bench.synthetic_documents(24000)

spent = {"validate": 0.0, "save_checkpoint": 0.0, "n_val": 0, "n_save": 0}
t_loop = {}


def timed(name, fn, count):
    def wrapper(*a, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        torch.cuda.synchronize()
        spent[name] += time.perf_counter() - t0
        spent[count] += 1
        return out
    return wrapper


prun.validate = timed("validate", prun.validate, "n_val")
prun.save_checkpoint = timed("save_checkpoint", prun.save_checkpoint, "n_save")
loop = prun.train_loop


def loop_timed(trainer, *a, **kw):
    if fp8:
        trainer.engine.set_fp8(True)
    torch.cuda.synchronize()
    t_loop["t0"] = time.perf_counter()
    out = loop(trainer, *a, **kw)
    torch.cuda.synchronize()
    t_loop["t1"] = time.perf_counter()
    return out


prun.train_loop = loop_timed
real_stdout = os.dup(1)
os.dup2(2, 1)            # the per-step records go to stderr here; the summary to stdout
torch.manual_seed(0)
pdata.seed_reference_streams(1)
t0 = time.perf_counter()
trainer, step, epoch = prun.train({"config_path": path, "run_name": "e2e"}, dataset=docs)
wall = time.perf_counter() - t0
os.dup2(real_stdout, 1)
run_dir = os.path.join(tmp, "runs", "e2e")
recs = [json.loads(l) for l in open(os.path.join(run_dir, "metrics.jsonl"))]
losses = [r["phoneme_loss"] for r in recs if "phoneme_loss" in r]
loop_s = t_loop["t1"] - t_loop["t0"]
train_s = loop_s - spent["validate"] - spent["save_checkpoint"]
tok = step * B * 512
print(json.dumps({
    "what": "plbert_amd.run.train end to end: worker processes -> pinned buffer -> copy stream -> device-side masking -> step, "
            "loss logged every step (read back one step late), validation + step_N.pth every save_interval",
    "dtype": "fp8" if fp8 else "bf16", "batch": B, "seq": 512, "steps": step, "epochs": epoch, "workers": workers,
    "save_interval": save_interval,
    "train_loop_s": round(loop_s, 2), "of_which_validation_s": round(spent["validate"], 2), "validation_passes": spent["n_val"],
    "of_which_checkpoints_s": round(spent["save_checkpoint"], 2), "checkpoints": spent["n_save"],
    "training_steps_ms_per_step": round(train_s / step * 1e3, 3),
    "training_steps_tokens_per_s": round(tok / train_s, 1),
    "whole_loop_tokens_per_s": round(tok / loop_s, 1),
    "train_call_wall_s": round(wall, 2),
    "records": len(recs), "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4),
    "files": sorted(f for f in os.listdir(run_dir) if f.startswith("step_")),
    "ln_exchange_timeouts": trainer.engine.status()["ln_exchange_timeouts"]}))
