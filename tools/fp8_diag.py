"""Which tensors carry the fp8 path's gradient error, and across which call transitions (GPU box).
   python tools/fp8_diag.py            (PLBERT_FP8_TN=0 for bf16 weight-gradient operands)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plbert_amd  # noqa: E402
from gpu_util import rel_l2  # noqa: E402
from plbert_amd.engine import HipEngine  # noqa: E402

pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                               max_position_embeddings=512, num_hidden_layers=12)
B, S = 32, 512
sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=3)
labels, masked, lengths, idx = plbert_amd.synthetic_batch(B, S, seed=99)


def make(fp8):
    eng = HipEngine(pcfg, 188, 0, max_batch=B, max_seq=S)
    eng.load_state_dict(sd)
    if fp8:
        eng.set_fp8(True)
    return eng


def run(eng, sel):
    off, flat = plbert_amd.masked_indices_to_csr([idx[i] for i in sel])
    loss = eng.loss_fwd_bwd(masked[sel], labels[sel], None, off, flat, int(off[-1]))
    torch.cuda.synchronize()
    return float(loss.item()), eng.grads[: eng.trainable].clone()


def report(tag, eng, g, ref):
    rows = []
    for name, (o, n, shp) in eng.layout.items():
        if o + n <= eng.trainable and float(ref[o:o + n].norm()) > 1e-9 and "key.bias" not in name:
            rows.append((rel_l2(g[o:o + n], ref[o:o + n]), name.split("albert_layers.0.")[-1], float(ref[o:o + n].norm())))
    rows.sort(reverse=True)
    print(f"--- {tag}: whole {rel_l2(g, ref):.3f}; worst: " + "; ".join(f"{n} {r:.3f}" for r, n, _ in rows[:6]), flush=True)


full = list(range(B))
perm = np.random.RandomState(1).permutation(B).tolist()
ref = make(False)
_, g_ref = run(ref, full)
_, g_ref_p = run(ref, perm)
print("bf16 perm vs full:", rel_l2(g_ref_p, g_ref))
q = B // 4
g_ref_q = [run(ref, list(range(k * q, k * q + q)))[1] for k in range(4)]
eng = make(True)
run(eng, full)
l, g = run(eng, full); report("fp8 call 2 (first fp8) vs bf16", eng, g, g_ref)
l, g1 = run(eng, full); report("fp8 call 3 vs bf16", eng, g1, g_ref)
for k in range(4):
    l, g = run(eng, list(range(k * q, k * q + q))); report(f"fp8 quarter {k} vs bf16 quarter", eng, g, g_ref_q[k])
l, g3 = run(eng, perm); report("fp8 perm (after quarters) vs bf16", eng, g3, g_ref)
report("fp8 perm vs fp8 call 3", eng, g3, g1)
l, g4 = run(eng, perm); report("fp8 perm again vs bf16", eng, g4, g_ref)
l, g5 = run(eng, full); report("fp8 full again vs bf16", eng, g5, g_ref)
