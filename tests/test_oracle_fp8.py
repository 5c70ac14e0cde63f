"""The fp8 restatement (oracle/fp8_np.py), on the CPU: its rounding against torch's OCP fp8 casts, its calibration
call against the pinned oracle (equal to 1e-10: the fp8 module adds rounding at the sites and nothing else), and the size of
the rounding's effect on a reference-captured fixture."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
from oracle import albert_np as onp
from oracle import fp8_np


@pytest.mark.parametrize("fmt,dt", [(fp8_np.E4M3, torch.float8_e4m3fn), (fp8_np.E5M2, torch.float8_e5m2)])
def test_rounding_matches_the_ocp_formats(fmt, dt):
    rs = np.random.RandomState(3)
    fmax = fmt["fmax"]
    x = np.concatenate([
        rs.randn(20000) * 3, rs.randn(20000) * fmax / 3, rs.randn(20000) * 2.0 ** (fmt["emin"] - 1),   # normals, large, subnormals
        np.ldexp(1.0, np.arange(fmt["emin"] - fmt["mant"] - 2, 9)),                                       # powers of two incl. below the smallest
        np.array([0.0, fmax, -fmax, fmax * 0.999, 2.0 ** (fmt["emin"] - fmt["mant"] - 1)]),             # tie at half the smallest subnormal -> 0
    ]).astype(np.float32)
    x = np.clip(x, -fmax, fmax)
    want = torch.from_numpy(x).to(dt).float().numpy()
    got = fp8_np.round_fp8(x, fmt)
    assert np.array_equal(got.astype(np.float32), want)
    # every tie between two neighbours goes to the even one
    grid = np.unique(want[np.isfinite(want)])
    mid = ((grid[:-1].astype(np.float64) + grid[1:]) / 2).astype(np.float32)
    assert np.array_equal(fp8_np.round_fp8(mid, fmt).astype(np.float32), torch.from_numpy(mid).to(dt).float().numpy())
    # saturation (the device clamps before converting: csrc/common.h pack_fp8x4)
    assert fp8_np.round_fp8(np.array([1e9, -1e9]), fmt).tolist() == [fmax, -fmax]


@pytest.mark.parametrize("name", ["tiny_h64", "small_h128"])
def test_calibration_call_is_the_pinned_oracle(name):
    g = load_golden(name)
    ocfg, _, sd = golden_cfg(g)
    idx = [list(map(int, x)) for x in g["index"]]
    lens = [int(x) for x in g["lengths"]]
    l0, p0, G0 = onp.loss_and_grads(ocfg, sd, g["masked"], g["labels"], lens, idx, dtype=np.float64)
    l1, p1, G1, amax = fp8_np.loss_and_grads_fp8(ocfg, sd, g["masked"], g["labels"], lens, idx, amax=None)
    # same arithmetic up to association (Q|K|V as one [3H, H] product, the softmax normalisation after the second product)
    assert abs(l1 - l0) < 1e-12 * l0 and np.allclose(p1, p0, rtol=1e-10, atol=1e-12)
    assert set(G1) == set(G0)
    for k in G0:
        assert np.abs(G1[k] - G0[k]).max() <= 1e-10 * np.abs(G0[k]).max() + 1e-18, k   # (key.bias: exactly 0 in exact arithmetic)
    assert set(amax) == set(fp8_np.ACT_SITES + fp8_np.GRAD_SITES) and all(v > 0 for v in amax.values())


def test_fp8_call_on_the_small_fixture():
    """The rounding moves the loss by < 1 % and the gradients by a visible but bounded amount (this 128-wide model is
    shallow and its gradient is mostly head and embeddings: ~1e-2; the 768/12 model on the device shows ~0.1 from its
    bf16 step, tests/test_gpu_fp8.py); with un-rounded operands for the weight gradients the distance shrinks."""
    g = load_golden("small_h128")
    ocfg, _, sd = golden_cfg(g)
    idx = [list(map(int, x)) for x in g["index"]]
    lens = [int(x) for x in g["lengths"]]
    args = (ocfg, sd, g["masked"], g["labels"], lens, idx)
    l0, _, G0, amax = fp8_np.loss_and_grads_fp8(*args)
    l8, _, G8, seen = fp8_np.loss_and_grads_fp8(*args, amax=amax)
    _, _, G8b, _ = fp8_np.loss_and_grads_fp8(*args, amax=amax, tn8=False)
    assert l8 != l0 and abs(l8 - l0) / l0 < 1e-2
    flat = lambda G: np.concatenate([G[k].reshape(-1) for k in sorted(G)])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    r, rb = rel(flat(G8), flat(G0)), rel(flat(G8b), flat(G0))
    assert 2e-3 < r < 0.25 and rb < r, (r, rb)
    # bf16 stores restated on top: finite, and a small move next to the fp8 rounding's
    l16, _, G16, _ = fp8_np.loss_and_grads_fp8(*args, amax=amax, bf16=True)
    assert abs(l16 - l8) / l8 < 1e-2 and rel(flat(G16), flat(G8)) < r
    # nothing left the formats' range: the activations were scaled to 448, the gradients to half of e5m2's range
    for s in fp8_np.ACT_SITES:
        assert seen[s] / amax[s] < 1.5
