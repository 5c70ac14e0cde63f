"""Does replaying the loss + backward call from a hipGraph beat launching it (config A, 768/12, 32 x 512)? The call is
capturable (tests/test_gpu_engine.py: bit-identical replay); this measures whether the ~300 launches per step cost anything
the graph would save. usage: python tools/graph_probe.py [bf16|fp8]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plbert_amd
from plbert_amd.train import PLBertTrainer

fp8 = len(sys.argv) > 1 and sys.argv[1] == "fp8"
cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=768, num_attention_heads=12,
                              intermediate_size=2048, max_position_embeddings=512, num_hidden_layers=12)
tr = PLBertTrainer(cfg, len(plbert_amd.symbols), max_batch=32, max_seq=512, lr=7e-5, seed=0)
if fp8:
    tr.engine.set_fp8(True)
b = tr.stage_batch(*plbert_amd.synthetic_batch(32, 512, seed=1234))
for _ in range(10):
    tr.step(b)
torch.cuda.synchronize()


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = lambda: tr.loss_and_grads(b)
t_eager = [timed(eager) for _ in range(3)]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.loss_and_grads(b)
t_graph = [timed(g.replay) for _ in range(3)]
t_eager2 = [timed(eager) for _ in range(3)]
print("loss + backward, ms per call:", "fp8" if fp8 else "bf16")
print("  launched :", [round(x, 3) for x in t_eager + t_eager2])
print("  replayed :", [round(x, 3) for x in t_graph])
