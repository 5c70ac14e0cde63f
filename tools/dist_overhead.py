"""What the gradient exchange costs beside the step on one GPU (world size 1 rehearsal):
   RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29540 python tools/dist_overhead.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plbert_amd  # noqa: E402
from plbert_amd.train import PLBertTrainer  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    cfg = plbert_amd.AlbertConfig(vocab_size=len(plbert_amd.symbols), hidden_size=768, num_attention_heads=12,
                                  intermediate_size=2048, max_position_embeddings=512, num_hidden_layers=12)
    tr = PLBertTrainer(cfg, len(plbert_amd.symbols), max_batch=32, max_seq=512, force_collectives=True)
    batch = tr.stage_batch(*plbert_amd.synthetic_batch(32, 512, seed=1234))
    eng = tr.engine
    g = eng.grads[: eng.trainable]
    side = torch.cuda.Stream()

    def none():
        pass

    def inline_one():
        dist.all_reduce(g)

    def side_one():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dist.all_reduce(g)
        main.wait_stream(side)

    def two_piece():  # a small early piece on the side stream + the rest, joined before AdamW
        main = torch.cuda.current_stream()
        h0 = eng.layout["phoneme_predictor.weight"][0]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dist.all_reduce(g[h0:])
        dist.all_reduce(g[:h0])
        main.wait_stream(side)

    def inline_async():
        w = dist.all_reduce(g, async_op=True)
        w.wait()

    for name, fn in (("no collective", none), ("one all_reduce, main stream", inline_one), ("one all_reduce, async_op", inline_async),
                     ("one all_reduce, side stream", side_one), ("two pieces, one on a side stream", two_piece), ("no collective", none)):
        for _ in range(3):
            tr.loss_and_grads(batch); fn(); tr.step_count += 1
            eng.adamw_step(tr.step_count, 7e-5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.loss_and_grads(batch); fn(); tr.step_count += 1
            eng.adamw_step(tr.step_count, 7e-5)
        torch.cuda.synchronize()
        print(f"{name:34s} {(time.perf_counter() - t0) * 100:.3f} ms/step", flush=True)
    for name, force in (("trainer.step, collectives on", True), ("trainer.step, collectives off", False),
                        ("trainer.step, collectives on", True)):
        tr.reducer.active = force
        for _ in range(5):
            tr.step(batch)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(batch)
        dist.barrier(); torch.cuda.synchronize()
        print(f"{name:34s} {(time.perf_counter() - t0) * 50:.3f} ms/step", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
