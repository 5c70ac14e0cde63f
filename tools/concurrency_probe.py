"""Does breaking the chip-wide lockstep pay? Two independent training steps of batch 16 on two HIP streams against
one step of batch 32: same tokens, same kernels at half the grid. If memory-bound phases of one stream overlap
MFMA-bound phases of the other, the pair finishes sooner than the single large step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plbert_amd
from plbert_amd.train import PLBertTrainer

cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                              max_position_embeddings=512, num_hidden_layers=12)
S = 512


def make(B, seed):
    tr = PLBertTrainer(cfg, 188, max_batch=B, max_seq=S, lr=7e-5, seed=0)
    lab, msk, lens, idx = plbert_amd.synthetic_batch(B, S, seed=seed)
    return tr, tr.stage_batch(lab, msk, lens, idx)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


big, bb = make(32, 1)
t32 = timeit(lambda: big.step(bb))
print(f"1 x batch 32: {t32:.3f} ms/step  ({32 * S / t32 * 1e3:.0f} tok/s)")
for nsplit in (2, 4):
    B = 32 // nsplit
    trs = [make(B, 10 + i) for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    t1 = timeit(lambda: trs[0][0].step(trs[0][1]))
    print(f"1 x batch {B} alone: {t1:.3f} ms/step ({B * S / t1 * 1e3:.0f} tok/s)")

    def both():
        for (tr, b), st in zip(trs, streams):
            with torch.cuda.stream(st):
                tr.step(b)

    tn = timeit(both)
    print(f"{nsplit} x batch {B} on {nsplit} streams: {tn:.3f} ms per {32 * S} tokens ({32 * S / tn * 1e3:.0f} tok/s) "
          f"vs {t32:.3f} ms: {t32 / tn:.3f}x")
    del trs
    torch.cuda.empty_cache()
