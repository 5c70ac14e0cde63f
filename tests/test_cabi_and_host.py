"""CPU checks of the boundary: the C-ABI library loads (against torch's HIP runtime) and exports
every symbol include/plbert.h declares; host logic that needs no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import plbert_amd
from plbert_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "plbert.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plb_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    decl = _declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(decl) == sorted(_lib.PUBLIC_SYMBOLS)
    for s in decl:
        assert hasattr(L, s), s


def test_one_hip_runtime_in_process():
    _lib.lib()
    maps = open("/proc/self/maps").read()
    libs = set(re.findall(r"(\S*libamdhip64\S*)", maps))
    assert len(libs) == 1, libs  # torch's; loading a second runtime would invalidate torch's stream handles


def test_host_only_calls_and_layout():
    """plb_create / plb_param_layout / plb_workspace_bytes touch no device."""
    L = _lib.lib()
    c = _lib.PlbConfig(188, 128, 768, 12, 2048, 12, 512, 2, 1e-12, 188, 0, 32, 512)
    h = C.c_void_p()
    assert L.plb_create(C.byref(c), C.byref(h)) == 0
    offs = (C.c_int64 * _lib.PLB_NPARAM)()
    sizes = (C.c_int64 * _lib.PLB_NPARAM)()
    total, train = C.c_int64(), C.c_int64()
    assert L.plb_param_layout(h, offs, sizes, C.byref(total), C.byref(train)) == 0
    assert total.value == 6438332                      # SURVEY.md §8(a) parameter inventory
    assert total.value - train.value == 590592         # the pooler, outside the AdamW range
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048)
    shapes = plbert_amd.param_shapes(cfg, 188)
    names = [n for n in _lib.PLB_PARAM_NAMES if n in shapes]
    assert list(shapes.keys()) == names
    for i, n in enumerate(_lib.PLB_PARAM_NAMES):
        if n in shapes:
            assert sizes[i] == int(np.prod(shapes[n])), n
            assert offs[i] % 4 == 0
    ws = L.plb_workspace_bytes(h)
    assert 4e9 < ws < 12e9                              # ~20 KB/token/layer of stash + grads, 16384 tokens
    L.plb_destroy(h)
    # rejected configurations fail loudly with a message
    bad = _lib.PlbConfig(188, 128, 768, 8, 2048, 12, 512, 2, 1e-12, 188, 0, 32, 512)
    assert L.plb_create(C.byref(bad), C.byref(h)) != 0
    assert b"head_dim" in L.plb_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        plbert_amd.HipEngine(cfg, 188)
    with pytest.raises(RuntimeError):
        plbert_amd.AlbertModel(cfg)


def test_config_yaml_surface(tmp_path):
    y = tmp_path / "config.yml"
    y.write_text("""
model_params:
  pretrained_model: ""
  hidden_size: 768
  num_attention_heads: 12
  intermediate_size: 2048
  max_position_embeddings: 512
  num_hidden_layers: 12
  dropout: 0.1
dataset_params:
  word_separator: 87
  max_seq_length: 512
  word_pred_prob: 0.15
  phoneme_mask_prob: 0.8
  replace_prob: 0.1
""")
    conf = plbert_amd.load_config(str(y))
    cfg = plbert_amd.albert_config_from_yaml(conf, vocab_size=len(plbert_amd.symbols))
    assert (cfg.vocab_size, cfg.embedding_size, cfg.hidden_size, cfg.num_hidden_layers) == (188, 128, 768, 12)
    assert cfg.dropout == 0.1 and cfg.pretrained_model == "" and cfg.hidden_dropout_prob == 0.0  # inert extras
    cfg.check_supported()
    with pytest.raises(ValueError):
        plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, hidden_act="gelu").check_supported()
    # shapes the engine has no kernels for are refused when the model is CONSTRUCTED (AlbertModel / HipEngine call
    # check_supported), with the reason in the message — not at the first launch
    for kw, word in ((dict(hidden_size=768, num_attention_heads=8), "head_dim"), (dict(hidden_size=2048, num_attention_heads=32), "1024"),
                     (dict(hidden_size=768, num_attention_heads=12, embedding_size=512), "embedding_size"),
                     (dict(hidden_size=768, num_attention_heads=12, intermediate_size=2000), "intermediate_size")):
        with pytest.raises(ValueError, match=word):
            plbert_amd.AlbertConfig(vocab_size=188, **kw).check_supported()
        with pytest.raises(ValueError, match=word):
            plbert_amd.AlbertModel(plbert_amd.AlbertConfig(vocab_size=188, **kw))


def test_validate_batch_rejects_bad_inputs():
    from plbert_amd.train import validate_batch

    lab = np.ones((2, 8), np.int64)
    validate_batch(lab, lab, [8, 5], [[0, 1], [4]], 188)
    with pytest.raises(ValueError):
        validate_batch(lab, lab, [8, 5], [[0, 1], [5]], 188)      # index beyond the length
    with pytest.raises(ValueError):
        validate_batch(lab, lab, [8, 5], [[1, 1], [0]], 188)      # duplicate
    with pytest.raises(ValueError):
        validate_batch(lab * 200, lab, [8, 5], [[0], [0]], 188)   # id outside the vocabulary
    with pytest.raises(ValueError):
        validate_batch(lab, lab, [9, 5], [[0], [0]], 188)         # length beyond S


def test_validate_token_ids_and_five_tuple_staging_checks():
    from plbert_amd.train import validate_token_ids

    tok = np.array([[3, 9, 0, 0], [1, 2, 5, 7]])
    validate_token_ids(tok, (2, 4), [2, 4], 10)                       # padding positions are not checked
    with pytest.raises(ValueError):
        validate_token_ids(tok, (2, 4), [2, 4], 9)                    # class id 9 needs num_tokens >= 10
    with pytest.raises(ValueError):
        validate_token_ids(tok[:, :3], (2, 4), [2, 4], 10)            # shape must match the phoneme batch
    bad = tok.copy()
    bad[1, 0] = -1
    with pytest.raises(ValueError):
        validate_token_ids(bad, (2, 4), [2, 4], 10)


def test_train_loop_reruns_a_step_the_engine_declared_invalid():
    """run._checked (the read-back of run.train_loop / run.validate): a step the engine declared invalid — HandoffTimeout from
    the status check right behind the loss read-back, where it is exact — is run again, at most MAX_HANDOFF_RETRIES times,
    and the retries are logged; anything else propagates. Host logic only: a stub stands in for the trainer."""
    from plbert_amd import run
    from plbert_amd.engine import HandoffTimeout

    class Loss:
        def __init__(self, v): self.v = v
        def item(self): return self.v

    class Engine:
        def __init__(self, fail): self.fail, self.checks = fail, 0
        def raise_if_failed(self):
            self.checks += 1
            if self.fail > 0:
                self.fail -= 1
                raise HandoffTimeout("injected")

    class Trainer:
        def __init__(self, fail): self.engine, self.calls = Engine(fail), 0
        def step(self):
            self.calls += 1
            return Loss(float("nan") if self.engine.fail > 0 else 1.25)

    notes = []
    tr = Trainer(fail=2)
    assert run._checked(tr, tr.step, lambda **kw: notes.append(kw)) == 1.25
    assert tr.calls == 3 and [n["retry"] for n in notes] == [1, 2] and all("injected" in n["handoff_timeout"] for n in notes)
    tr = Trainer(fail=run.MAX_HANDOFF_RETRIES + 1)
    with pytest.raises(HandoffTimeout):
        run._checked(tr, tr.step)
    assert tr.calls == run.MAX_HANDOFF_RETRIES + 1
    tr = Trainer(fail=0)

    def boom():
        raise ValueError("not a hand-off matter")
    with pytest.raises(ValueError):
        run._checked(tr, boom)


@pytest.mark.parametrize("deferred", [True, False])
@pytest.mark.parametrize("fail_at", [(), (0,), (2,), (3,), (5,), (1, 4)])
def test_train_loop_with_deferred_readback_applies_every_batch_once_in_order(monkeypatch, deferred, fail_at):
    """run.train_loop reads the loss of step i back after step i + 1 has been enqueued (no stall). Host logic against a
    simulated device: a step the engine declares invalid is left out on the 'device' together with everything enqueued
    behind it until the host has read the status (the sticky word), and the loop must end with EVERY batch applied exactly
    once, in order, the records 1..N in order, checkpoints / validation exactly at the interval boundaries and never a step
    enqueued beyond num_steps — whichever step fails (first, last of an interval, last of the run, two of them)."""
    from plbert_amd import run
    from plbert_amd.engine import HandoffTimeout

    class Device:
        def __init__(self):
            self.applied, self.sticky, self.skipped, self.enqueued = [], False, 0, 0

    dev = Device()

    class Engine:
        device = "sim"

        def raise_if_failed(self):
            if dev.sticky:
                n, dev.sticky, dev.skipped = dev.skipped, False, 0
                err = HandoffTimeout(f"{n} skipped")
                err.skipped_updates = n
                raise err

    class Trainer:
        engine = Engine()

        def step(self, batch):
            self.engine.raise_if_failed()                   # what HipEngine._loss_call does first
            n, dev.enqueued = dev.enqueued, dev.enqueued + 1
            if n in fail_at or dev.sticky:
                dev.sticky, dev.skipped = True, dev.skipped + 1
                return _Loss(float("nan"))
            dev.applied.append(batch)
            return _Loss(float(batch))

    class _Loss(float):
        def item(self): return float(self)

    class Reader:
        def post(self, loss): return loss
        def read(self, h): return float(h)

    N, interval = 6, 3
    events = []
    monkeypatch.setattr(run, "_batches", lambda *a, **k: iter(range(100)))
    monkeypatch.setattr(run, "validate", lambda *a, **k: (events.append(("validate", len(dev.applied), dev.enqueued)), 0.5)[1])
    monkeypatch.setattr(run, "save_checkpoint", lambda tr, step, *a: events.append(("save", step, list(dev.applied))))
    grants = []

    class Budget:                                                        # stands in for the training loader's DeviceFeeder
        def grant(self, n): grants.append(n)
        def reset_budget(self, n=0): grants.append(("reset", n))

    monkeypatch.setattr(run, "_source", lambda loader, trainer: loader)
    monkeypatch.setattr(run, "_feeder", lambda *a, **k: Budget())
    recs = []
    step, epoch = run.train_loop(Trainer(), None, None, 0, N, interval, 2, lambda **kw: recs.append(kw), "unused",
                                 deferred_readback=deferred, reader=Reader())
    assert step == N and dev.applied == list(range(N))                 # every batch once, in order, none beyond num_steps
    assert dev.enqueued >= N + len(fail_at)                             # (each invalid step and what was enqueued behind it ran again)
    losses = [r for r in recs if "phoneme_loss" in r]
    assert [r["step"] for r in losses] == list(range(1, N + 1)) and [r["phoneme_loss"] for r in losses] == [float(i) for i in range(N)]
    saves = [e for e in events if e[0] == "save"]
    assert [(e[1], e[2]) for e in saves] == [(3, [0, 1, 2]), (6, [0, 1, 2, 3, 4, 5])]      # checkpoints on the reference's weights
    vals = [e for e in events if e[0] == "validate"]
    assert [e[1] for e in vals] == [0, 3, 6] and all(e[1] == len([b for b in range(e[1])]) for e in vals)
    assert len([r for r in recs if "handoff_timeout" in r]) >= len(fail_at) > 0 or not fail_at
    assert grants == [("reset", 0), 3, 3, 3]                             # draws granted interval by interval (start, step 3, step 6)


def test_bench_maps_profiler_classes_to_rocprof_kernel_names():
    """bench.py attributes PMC traffic to the dominant profiler class by kernel name (template arguments
    <tile, ACT, OUTF32, loop form, FP8, ABF8> of the pipeline GEMM, <ACT, OUTF32> of the 128x128 one, which is a class
    of its own: some of its launches queue on the side stream)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    big = "void (anonymous namespace)::gemm_nt_big_kernel<%d, %d, %s, true, false, false>(PlbGemmNT)"
    assert bench._in_class("gemm_nt", "void (anonymous namespace)::gemm_nt_big_kernel<3, 0, false, true>(PlbGemmNT)")  # round-1 names
    assert bench._in_class("gemm_nt", "void (anonymous namespace)::gemm_nt_big_kernel<3, 0, false, true, true, true>(PlbGemmNT)")
    assert bench._in_class("gemm_nt", big % (3, 0, "false"))
    assert not bench._in_class("gemm_nt", big % (2, 1, "false"))
    assert bench._in_class("gemm_nt_gelu", big % (2, 1, "false"))
    assert bench._in_class("gemm_nt_gelubwd", big % (2, 2, "false"))
    assert bench._in_class("gemm_nt_f32", big % (2, 0, "true")) and not bench._in_class("gemm_nt", big % (2, 0, "true"))
    small = "void (anonymous namespace)::gemm_nt_kernel<0, false>(PlbGemmNT)"
    assert bench._in_class("gemm_nt_small", small) and not bench._in_class("gemm_nt", small)
    assert bench._in_class("gemm_nt_f32", "void (anonymous namespace)::gemm_nt_kernel<0, true>(PlbGemmNT)")
    assert not bench._in_class("gemm_nt_small", big % (3, 0, "false"))
    assert bench._in_class("gemm_tn", "(anonymous namespace)::gemm_tn_big_kernel(PlbGemmTN)")
    assert not bench._in_class("gemm_nt", "(anonymous namespace)::gemm_tn_big_kernel(PlbGemmTN)")
    assert bench._in_class("attn_bwd_dkv", "(anonymous namespace)::attn_bwd_dkv_kernel(PlbAttn)")


def test_bench_parent_spawns_ranks_without_touching_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` as typed: the parent must start N ranks through torch.distributed.run on
    127.0.0.1 BEFORE importing torch / making any GPU call, and relay rank 0's single JSON line."""
    import importlib
    import subprocess
    import sys as _sys
    import types

    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "phoneme-tokens/sec", "n_gpus": 4}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("RANK", raising=False)
    torch_loaded_before = "torch.cuda" in _sys.modules and _sys.modules["torch"].cuda.is_initialized()
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "phoneme-tokens/sec", "n_gpus": 4}'
    if "torch" in _sys.modules:
        assert _sys.modules["torch"].cuda.is_initialized() == torch_loaded_before


def _disassemble_code_objects(tmp_path):
    """Every gfx950 code object of the in-tree library, disassembled (llvm-objdump ships with ROCm; --offloading
    writes the bundles next to its input, so it runs on a copy)."""
    import glob
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    lib = shutil.copy(os.path.join(ROOT, "plbert_amd", "libplbert_hip.so"), str(tmp_path))
    subprocess.run([objdump, "--offloading", lib], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    cos = sorted(glob.glob(lib + ".*gfx950"))
    assert cos, "no gfx950 code object extracted"
    for co in cos:
        yield co, subprocess.run([objdump, "-d", "--no-show-raw-insn", co], check=True, stdout=subprocess.PIPE,
                                 text=True).stdout


def test_m0_is_only_touched_by_the_dma_statements(tmp_path):
    """plbert_amd/build.py: verify_m0 — the post-link check every build runs (see there): in kernels that contain the asm
    LDS-DMA statements of csrc/common.h, M0 is touched by those statements only; no kernel ever reads M0."""
    from plbert_amd import build as plb_build

    if not os.path.exists(plb_build.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    n_asm = plb_build.verify_m0(os.path.join(ROOT, "plbert_amd", "libplbert_hip.so"), str(tmp_path))
    assert n_asm > 1000  # the pipeline GEMMs and the attention kernels really are LDS-DMA kernels


def test_no_kernel_of_the_step_uses_scratch(tmp_path):
    """Register spills go to scratch memory (HBM round trips inside the hottest loops): the code objects' metadata must
    show none, for EVERY kernel of the library (round 5: the single-kernel attention backward, which runs at the full
    512-register file, lost its 21 spilled registers — lane constants of rare paths and of the epilogue are re-derived
    where they are used instead of being carried across its loop)."""
    import subprocess

    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    seen = 0
    for co, _ in _disassemble_code_objects(tmp_path):
        notes = subprocess.run([readelf, "--notes", co], check=True, stdout=subprocess.PIPE, text=True).stdout
        for blk in notes.split(".name:")[1:]:
            name = blk.split()[0]
            m = re.search(r"\.vgpr_spill_count:\s*(\d+)", blk)
            q = re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk)
            if not m or not q:
                continue
            seen += 1
            spills, scratch = int(m.group(1)), int(q.group(1))
            assert spills == 0 and scratch == 0, (name, spills, scratch)
    assert seen > 40


def test_library_compiles_without_warnings(tmp_path):
    """-Wall build of every source: a warning is either a real problem or noise that hides one."""
    import subprocess
    from plbert_amd import build as B

    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not found")
    procs = []
    for src in B.SOURCES:
        cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o",
               str(tmp_path / (src + ".o"))]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out[-2000:]
        assert "warning" not in out, f"{src}:\n{out[:3000]}"


def test_fp8_mfmas_stay_in_their_hand_placed_slots(tmp_path):
    """The interleaved K loops place ONE MFMA per slot and deal LDS reads / DMA issues into its shadow; sched_barrier pins
    that order for the machine scheduler, but LLVM's MachineSink pass runs later and — in the fp8 builds only (8-register
    operand tuples) — sank whole phases of MFMAs into the loop latch (runs of 16-23 back to back, +15-30 VGPRs). build.py
    compiles the fp8 translation units with that pass off; checked here on the assembly of the weight-gradient kernel:
    between two sched_barrier markers there is never more than one MFMA."""
    import subprocess

    from plbert_amd import build as plb_build

    src = os.path.join(ROOT, "plbert_amd", "csrc", "gemm_tn_fp8.hip")
    out = str(tmp_path / "tn8.s")
    cmd = [plb_build._hipcc(), *plb_build.FLAGS, *plb_build.EXTRA_FLAGS.get("gemm_tn_fp8.hip", []), "-x", "hip", "-S",
           "--cuda-device-only", src, "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    run = best = total = 0
    for ln in open(out):
        t = ln.strip()
        if not t or t.startswith(";"):
            if "sched_barrier" in t:
                run = 0
            continue
        if t.startswith("v_mfma"):
            run += 1
            total += 1
            best = max(best, run)
        else:
            run = 0
    assert total == 64 and best == 1, (total, best)   # two K-tiles of 32 MFMAs, each alone in its slot
