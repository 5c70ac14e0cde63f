"""Do the fp8 and the bf16 step train alike? 768/12 model, batch 32 x 512, N steps over a cycle of K distinct synthetic
batches (random data: only memorisation lowers the loss, which falls from ln(188)), both modes from the same initial
weights and the same batches. Prints the two loss curves at checkpoints and their largest relative distance.
   python tools/fp8_convergence.py [steps] [lr]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plbert_amd  # noqa: E402
from plbert_amd.train import PLBertTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4
cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                              max_position_embeddings=512, num_hidden_layers=12)
K = 8
curves = {}
for mode in ("bf16", "fp8"):
    tr = PLBertTrainer(cfg, 188, max_batch=32, max_seq=512, lr=lr, seed=7)
    if mode == "fp8":
        tr.engine.set_fp8(True)
    batches = [tr.stage_batch(*plbert_amd.synthetic_batch(32, 512, seed=100 + k)) for k in range(K)]
    losses = []
    for i in range(steps):
        losses.append(tr.step(batches[i % K]).clone())   # step() returns the engine's loss buffer: copy it (no host sync)
    torch.cuda.synchronize()
    curves[mode] = np.array([float(x.item()) for x in losses])
    st = tr.engine.status()
    if mode == "fp8":
        print("fp8 sites (calls in which values were clamped under the delayed scale, worst overshoot):", tr.engine.fp8_stats(), flush=True)
    print(f"{mode}: ln_exchange_timeouts {st['ln_exchange_timeouts']}, all finite {bool(np.isfinite(curves[mode]).all())}, "
          f"params finite {bool(torch.isfinite(tr.engine.params).all().item())}", flush=True)
    del tr, batches
b, f = curves["bf16"], curves["fp8"]
print(f"{steps} steps, lr {lr}, {K} batches of 32 x 512 in a cycle; loss (mean over the cycle ending at the step):")
print(" step    bf16     fp8    rel.diff")
for i in range(K - 1, steps, max(K, steps // 16 // K * K)):
    mb, mf = b[i - K + 1:i + 1].mean(), f[i - K + 1:i + 1].mean()
    print(f"{i + 1:5d}  {mb:7.4f} {mf:7.4f}  {abs(mf - mb) / mb:8.4f}")
w = K
rb = np.convolve(b, np.ones(w) / w, mode="valid")
rf = np.convolve(f, np.ones(w) / w, mode="valid")
print(f"largest relative distance of the cycle-averaged curves: {np.max(np.abs(rf - rb) / rb):.4f}; final {rb[-1]:.4f} (bf16) vs {rf[-1]:.4f} (fp8)")
