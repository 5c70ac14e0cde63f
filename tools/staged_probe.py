"""Where does the per-step staging cost go? python tools/staged_probe.py  (GPU box)
   wall per step of: the resident batch, the staged feeder, the feeder without its copies, and the host time to ENQUEUE a
   step in each form (no device wait)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plbert_amd  # noqa: E402
from plbert_amd.train import PLBertTrainer  # noqa: E402

B, S, K = 32, 512, 40
cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=768, num_attention_heads=12,
                              intermediate_size=2048, max_position_embeddings=512, num_hidden_layers=12)
tr = PLBertTrainer(cfg, num_phonemes=len(plbert_amd.symbols), max_batch=B, max_seq=S, lr=7e-5, device="cuda:0", seed=0)
labels, masked, lengths, idx = plbert_amd.synthetic_batch(B, S, seed=1234)
batch = tr.stage_batch(labels, masked, lengths, idx)
feeder = bench.StagedFeeder(tr, 16, B, S, 99, torch)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3


def resident():
    for _ in range(K):
        tr.step(batch)


def staged():
    feeder.run(K)


def staged_no_copy():
    for i in range(K):
        tr.step(feeder.batch(i))


for name, fn in (("resident", resident), ("staged", staged), ("staged, no copies", staged_no_copy), ("resident", resident),
                 ("staged", staged)):
    fn()
    enq, wall = timed(fn)
    print(f"{name:20s} enqueue {enq:6.3f} ms/step   wall {wall:6.3f} ms/step", flush=True)
