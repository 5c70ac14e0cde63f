#!/usr/bin/env python
"""PL-BERT pre-training hot path on MI355X: phoneme-tokens/s of the full training step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = masked-phoneme forward -> per-sample masked cross-entropy -> backward -> gradient
all-reduce (N > 1, RCCL) -> AdamW, on one synthetic fixed-length batch per rank that is resident in
HBM before the timed region (BASELINE.json configs[1]: ALBERT hidden 768 / 12 shared layers / FFN
2048, seq 512, batch 32 per GPU; weak scaling).  Prints ONE JSON line on rank 0.

roofline: the dominant kernel class of the step, algorithmic FLOPs per launch / average launch
duration, from per-launch HIP events recorded on the launch stream during a re-run of the same K
steps (the clean timed region carries no events).  cpu_baseline: the numpy oracle's training step
(oracle/albert_np.py, kind "port") on the host cores, rank 0 at N=1 only, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FLOP_PER_TOKEN_STEP = 454_440_960  # 3 x 151,480,320 forward, config A


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline sample")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that fill roofline.traffic (N = 1 only)")
    ap.add_argument("--model", choices=["base", "large"], default="base",
                    help="base = BASELINE configs[1] (hidden 768 / 12 layers, the bench line); large = configs[3] "
                         "(hidden 1024 / 24 layers / 16 heads / FFN 4096, batch 16): utilisation report only")
    ap.add_argument("--num-tokens", type=int, default=0,
                    help="> 0: dual-head step (MultiTaskModel: phoneme loss + grapheme/token loss over a vocabulary "
                         "of this size, e.g. 64000) instead of the reference's phoneme-only step; a second, "
                         "separately reported workload")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gradient all-reduce even at world size 1 "
                         "(rehearses the N>1 code path on a one-GPU box)")
    return ap.parse_args()


def cpu_baseline(budget_s):
    """The oracle's full training step (fwd + loss + bwd + AdamW, fp32 numpy) on a bounded sample of
    the same workload: batches of 2 x 512 tokens, repeated until the budget is used."""
    from threadpoolctl import threadpool_limits
    from oracle import albert_np as onp
    import plbert_amd

    threads = min(os.cpu_count() or 1, 16)
    pcfg = plbert_amd.AlbertConfig(vocab_size=188, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                                   max_position_embeddings=512, num_hidden_layers=12)
    sd = plbert_amd.deterministic_state_dict(pcfg, 188, seed=0)
    ocfg = onp.Config()
    B, S = 2, 512
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(B, S, seed=1234)
    opt = onp.AdamW(lr=7e-5)
    with threadpool_limits(limits=threads):
        onp.train_step(ocfg, sd, opt, masked, labels, lengths, idx)  # warm-up (BLAS init, page faults)
        t0 = time.perf_counter()
        n = 0
        while True:
            onp.train_step(ocfg, sd, opt, masked, labels, lengths, idx)
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    return {"value": round(n * B * S / dt, 1), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{n} full training steps (fwd+loss+bwd+AdamW, fp32 numpy oracle) of {B}x{S} tokens, "
                      f"768/12 model, {dt:.1f} s wall"}


# profiler class -> does a rocprof kernel name belong to it (template arguments: <tile, ACT, OUTF32, loop form>)
def _in_class(cls, name):
    import re
    if cls in ("gemm_nt", "gemm_nt_gelu", "gemm_nt_gelubwd", "gemm_nt_f32"):
        m = re.search(r"gemm_nt_big_kernel<\d+, (\d+), (false|true), (?:false|true)>", name) or \
            re.search(r"gemm_nt_kernel<(\d+), (false|true)>", name)
        if not m:
            return False
        act, f32 = int(m.group(1)), m.group(2) == "true"
        return {"gemm_nt": act == 0 and not f32, "gemm_nt_gelu": act == 1, "gemm_nt_gelubwd": act == 2,
                "gemm_nt_f32": f32}[cls]
    return cls.replace("_fwd", "").replace("_bwd", "") in name or cls in name


def pmc_traffic(cls):
    """HBM bytes per launch of the dominant kernel class from PMC counters, as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kernel trace only beside them) over a short
    child run of this same script; FETCH_SIZE is in KiB and counts a wide streaming read at half its bytes on
    gfx950 (x 1024 x 2), WRITE_SIZE in KiB (x 1024). Returns None when the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3"):
        return None
    tot, launches = {}, 0
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="plb_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-roofline", "--no-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            s, n = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == ctr and _in_class(cls, row["Kernel_Name"]):
                    s += float(row["Counter_Value"])
                    n += 1
            if n == 0:
                return None
            tot[ctr] = s / n
            launches = n
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
    return {"bytes": round(fetch + write), "fetch": round(fetch), "write": round(write), "launches_sampled": launches}


def main():
    args = parse()
    # The contract is ONE JSON line on stdout. Native libraries write there too (RCCL prints a version banner
    # when the first communicator is created), so fd 1 points at stderr until the line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    import plbert_amd
    from plbert_amd import _lib
    from plbert_amd.train import PLBertTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run
    if world > 1 or (launched and args.force_dist):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # nccl == RCCL on ROCm. No device_id: that would create the communicator NOW, and device buffers
        # allocated after an RCCL communicator exists are slower to use on this stack (measured at world size
        # 1: 11.3 instead of 10.85 ms/step, with or without collectives in the step). Lazily, the communicator
        # appears at the first collective — the start-up parameter broadcast, after the engine's allocations.
        dist.init_process_group("nccl")

    if args.model == "large":
        cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=1024, num_attention_heads=16,
                                      intermediate_size=4096, max_position_embeddings=512, num_hidden_layers=24)
        flop_per_token = 1_964_875_776
        if args.batch == 32:
            args.batch = 16
        model_desc = "hidden 1024 / 24 shared layers / FFN 4096 / 16 heads"
    else:
        cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=768, num_attention_heads=12,
                                      intermediate_size=2048, max_position_embeddings=512, num_hidden_layers=12)
        flop_per_token = FLOP_PER_TOKEN_STEP
        model_desc = "hidden 768 / 12 shared layers / FFN 2048 / 12 heads"
    B, S = args.batch, args.seq
    trainer = PLBertTrainer(cfg, num_phonemes=len(plbert_amd.symbols), max_batch=B, max_seq=S, lr=7e-5,
                            device=f"cuda:{local_rank}", seed=0, force_collectives=args.force_dist,
                            num_tokens=args.num_tokens)
    labels, masked, lengths, idx = plbert_amd.synthetic_batch(B, S, seed=1234 + rank)
    token_ids = None
    if args.num_tokens:
        token_ids = np.random.RandomState(4321 + rank).randint(0, args.num_tokens, size=(B, S)).astype(np.int64)
        flop_per_token += 3 * 2 * cfg.hidden_size * args.num_tokens  # token head fwd + dgrad + wgrad
        model_desc += f" + token head {args.num_tokens}"
    batch = trainer.stage_batch(labels, masked, lengths, idx, token_ids=token_ids)  # resident in HBM before timing

    def sync_all():
        if dist.is_initialized():
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    if dist.is_initialized():
        # Creating the RCCL communicator (the start-up broadcast above is the first collective) leaves the
        # process slow for some hundred milliseconds — measured: 12.4 instead of 10.8 ms/step over the 25 steps
        # that follow it, at world size 1 — which has nothing to do with the steady state being measured.
        # Settle untimed (every collective type the loop uses has then run once, too), then do the W warm-up steps.
        # A FIXED number of steps: every step holds a collective, so all ranks must run the same count.
        sync_all()
        for _ in range(120):
            trainer.step(batch)
        sync_all()
    for _ in range(args.warmup):
        trainer.step(batch)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())

    roofline = None
    if not args.no_roofline:
        _lib.profile_enable(True)
        for _ in range(args.steps):
            trainer.step(batch)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        if rank == 0 and prof:
            total_ms = sum(v["ms"] for v in prof.values())
            name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
            per_launch_flop = dom["flops"] / dom["launches"]
            avg_ms = dom["ms"] / dom["launches"]
            ach = per_launch_flop / (avg_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                        "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": dom["launches"] // args.steps,
                        "share_of_kernel_time": round(dom["ms"] / total_ms, 3),
                        "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in
                                               sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                        "_algo_bytes": round(dom["bytes"] / dom["launches"]),
                        "step_mfma_frac": round(flop_per_token * B * S / (total_ms / args.steps * 1e-3) / 1e12
                                                / MFMA_BF16_PEAK_TFLOPS, 4)}

    if roofline is not None and world == 1 and not dist.is_initialized() and not args.no_traffic:
        t = pmc_traffic(roofline["kernel"])
        if t is not None:
            roofline["traffic"] = t["bytes"]
            roofline["traffic_detail"] = {"unit": "B per launch, PMC", "fetch(FETCH_SIZE*1024*2)": t["fetch"],
                                          "write(WRITE_SIZE*1024)": t["write"], "launches_sampled": t["launches_sampled"],
                                          "algorithmic": roofline.pop("_algo_bytes", None)}
    if roofline is not None:
        roofline.pop("_algo_bytes", None)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.num_tokens:
        cpu = cpu_baseline(args.cpu_seconds)

    if rank == 0:
        tokens = world * B * S * args.steps
        out = {
            "metric": "phoneme-tokens/sec", "value": round(tokens / dt, 1), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"PL-BERT {'dual-head (phoneme + token loss)' if args.num_tokens else 'masked-phoneme'} "
                                   f"training step (fwd+loss+bwd+allreduce+AdamW), ALBERT "
                                   f"{model_desc}, seq_len {S}, batch {B} per GPU",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}"},
            "step_loss": round(loss_val, 5),
            "step_mfma_frac_wall": round(flop_per_token * B * S / (dt / args.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
