"""Stability soak: N training steps on two fixed synthetic batches (random data: only memorisation can lower the
loss); the loss must stay finite and fall, the parameters stay finite.
usage: soak.py [steps] [bf16|fp8] [base|large]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plbert_amd
from plbert_amd.train import PLBertTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fp8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
large = len(sys.argv) > 3 and sys.argv[3] == "large"
B = 16 if large else 32
cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=1024 if large else 768,
                              num_attention_heads=16 if large else 12, intermediate_size=4096 if large else 2048,
                              max_position_embeddings=512, num_hidden_layers=24 if large else 12)
tr = PLBertTrainer(cfg, len(plbert_amd.symbols), max_batch=B, max_seq=512, lr=1e-4)
if fp8:
    tr.engine.set_fp8(True)
print("soak:", "fp8" if fp8 else "bf16", "1024/24, 16 x 512" if large else "768/12, 32 x 512", flush=True)
batches = [tr.stage_batch(*plbert_amd.synthetic_batch(B, 512, seed=100 + i)) for i in range(2)]
losses = []
t0 = time.perf_counter()
for i in range(steps):
    losses.append(tr.step(batches[i % 2]).clone())   # step() returns the engine's loss buffer: copy it
    if (i + 1) % 5000 == 0:
        print(f"  step {i + 1}: loss {float(losses[-1].item()):.4f}", flush=True)
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
if fp8:
    print("fp8 sites (calls with clamped values, worst overshoot):", tr.engine.fp8_stats())
