"""Stability soak: N training steps on two fixed synthetic batches (random data: only memorisation can lower the
loss); the loss must stay finite and fall, the parameters stay finite."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plbert_amd
from plbert_amd.train import PLBertTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=768, num_attention_heads=12,
                              intermediate_size=2048, max_position_embeddings=512, num_hidden_layers=12)
tr = PLBertTrainer(cfg, len(plbert_amd.symbols), max_batch=32, max_seq=512, lr=1e-4)
batches = [tr.stage_batch(*plbert_amd.synthetic_batch(32, 512, seed=100 + i)) for i in range(2)]
losses = []
t0 = time.perf_counter()
for i in range(steps):
    losses.append(tr.step(batches[i % 2]).clone())   # step() returns the engine's loss buffer: copy it
torch.cuda.synchronize()
dt = time.perf_counter() - t0
l = torch.stack([x.reshape(()) for x in losses]).cpu().numpy() if False else np.array([float(x.item()) for x in losses])
print(f"{steps} steps in {dt:.2f} s ({dt/steps*1e3:.2f} ms/step incl. host reads at the end)")
print("loss first 5:", np.round(l[:5], 4), "last 5:", np.round(l[-5:], 4), "finite:", bool(np.isfinite(l).all()))
assert np.isfinite(l).all() and l[-20:].mean() < l[:20].mean()
p = tr.engine.params
print("params finite:", bool(torch.isfinite(p).all().item()), "max |p| %.3f" % float(p.abs().max().item()))
st = tr.engine.status() if hasattr(tr.engine, "status") else None
print("in-launch hand-off time-outs (plb_status):", st)
assert st is None or st["ln_exchange_timeouts"] == 0
