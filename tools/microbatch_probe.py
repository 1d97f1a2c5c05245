#!/usr/bin/env python3
"""EXPERIMENT: how much of the step's idle time (small decoder / ConvLSTM launches that leave most of the chip empty)
could two half-batches running on two streams fill?  Two independent trainers (own model, own plan, own graph) with
batch B/2 each are replayed on two streams and their aggregate rate is compared with one trainer at batch B.

    python tools/microbatch_probe.py [--B 32] [--steps 30]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climate_amd.config import synthetic_config  # noqa: E402
from climate_amd.model import get_model  # noqa: E402
from climate_amd.trainer import HotPathTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = synthetic_config(base_channels=32, seq_len=6)


def make(b):
    torch.manual_seed(42)
    m = get_model(cfg).to(dev)
    tr = HotPathTrainer(m, lr=cfg.training.lr, use_graph=True, distributed=False)
    x = torch.randn(b, 6, 5, 48, 72, device=dev)
    y = torch.randn(b, 2, 48, 72, device=dev)
    sx, sy = tr.input_buffers(x.shape, y.shape)
    sx.copy_(x); sy.copy_(y)
    return tr, sx, sy


def rate(trs, streams, steps):
    for _ in range(5):
        for (tr, x, y), s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (tr, x, y), s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(x.shape[0] for _, x, _ in trs)
    return n * steps / dt, dt / steps * 1e3


one = [make(a.B)]
r, ms = rate(one, [torch.cuda.current_stream()], a.steps)
print(f"one trainer, batch {a.B}: {r:8.1f} samples/s  {ms:.3f} ms/step", flush=True)
half = [make(a.B // a.parts)]
r, ms = rate(half, [torch.cuda.current_stream()], a.steps)
print(f"one trainer, batch {a.B // a.parts}: {r:8.1f} samples/s  {ms:.3f} ms/step", flush=True)
parts = half + [make(a.B // a.parts) for _ in range(a.parts - 1)]
streams = [torch.cuda.Stream() for _ in parts]
r, ms = rate(parts, streams, a.steps)
print(f"{a.parts} trainers, batch {a.B // a.parts} each, {a.parts} streams: {r:8.1f} samples/s  {ms:.3f} ms per pair of steps",
      flush=True)
r, ms = rate(parts, [torch.cuda.current_stream()] * len(parts), a.steps)
print(f"{a.parts} trainers, batch {a.B // a.parts} each, ONE stream: {r:8.1f} samples/s  {ms:.3f} ms per pair of steps",
      flush=True)
