"""Does training with the last application evaluated on the masked rows only (the default) follow training with every row
evaluated (PLBERT_PRUNE_LAST=0)? Same initial weights, same cycle of K synthetic batches (32 x 512, 768/12), N steps each;
both are bf16 evaluations of the same function, so the curves may differ only as two bf16 runs do.
   python tools/prune_convergence.py [steps] [lr]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plbert_amd  # noqa: E402
from plbert_amd import _lib  # noqa: E402
from plbert_amd.train import PLBertTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4
cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                              max_position_embeddings=512, num_hidden_layers=12)
K = 8
curves = {}
L = _lib.lib()
for mode in (0, 1):
    L.plb_set_prune_last(mode)
    tr = PLBertTrainer(cfg, 188, max_batch=32, max_seq=512, lr=lr, seed=7)
    batches = [tr.stage_batch(*plbert_amd.synthetic_batch(32, 512, seed=100 + k)) for k in range(K)]
    losses = [tr.step(batches[i % K]).clone() for i in range(steps)]
    torch.cuda.synchronize()
    curves[mode] = np.array([float(x.item()) for x in losses])
    print(f"{'pruned' if mode else 'every row'}: last_application_rows {tr.engine.last_application_rows()}, time-outs "
          f"{tr.engine.status()['ln_exchange_timeouts']}, finite {bool(np.isfinite(curves[mode]).all())}", flush=True)
    del tr, batches
L.plb_set_prune_last(-1)
a, b = curves[0], curves[1]
print(f"{steps} steps, lr {lr}, {K} batches in a cycle; loss (mean over the cycle ending at the step):")
print(" step  every row   pruned   rel.diff")
for i in range(K - 1, steps, max(K, steps // 16 // K * K)):
    ma, mb = a[i - K + 1:i + 1].mean(), b[i - K + 1:i + 1].mean()
    print(f"{i + 1:5d}  {ma:9.4f} {mb:8.4f}  {abs(mb - ma) / ma:8.5f}")
ra, rb = np.convolve(a, np.ones(K) / K, mode="valid"), np.convolve(b, np.ones(K) / K, mode="valid")
print(f"first-step losses {a[0]:.6f} vs {b[0]:.6f}; largest relative distance of the cycle-averaged curves {np.max(np.abs(rb - ra) / ra):.5f}; "
      f"final {ra[-1]:.4f} vs {rb[-1]:.4f}")
