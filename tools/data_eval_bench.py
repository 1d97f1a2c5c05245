#!/usr/bin/env python3
"""Measurement for the callers either side of the hot path at the reference's sizes: the device
window builder (8109 training time steps, batch 32 x 6 frames) and the device evaluator (1080 validation steps), with the
reference's own host-side way timed beside them on this box's cores (python loop + torch.stack + pinned H2D;
numpy inverse transform + moment sums)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from climate_amd.data import DeviceLoader, DeviceWindowDataset  # noqa: E402
from climate_amd.evaluation import DeviceEvaluator  # noqa: E402

N, C, H, W, T, B = 8109, 5, 48, 72, 6, 32
g = torch.Generator().manual_seed(0)
inp = torch.randn(N, C, H, W, generator=g); out = torch.randn(N, 2, H, W, generator=g)
ds = DeviceWindowDataset(inp, out, T)
x = torch.empty(B, T, C, H, W, device="cuda"); y = torch.empty(B, 2, H, W, device="cuda")
idx = torch.randint(0, N, (B,), generator=g).cuda()
for _ in range(3):
    ds.batch_into(idx, x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    ds.batch_into(idx, x, y)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
byts = (x.numel() + y.numel()) * 4 * 2          # read + write
print(f"cm_build_windows  batch {B} x {T} frames: {us:7.1f} us  ({byts / us / 1e3:6.1f} GB/s read+write; {B / us * 1e6:9.0f} samples/s)")
# a whole epoch through the loader (sampler on the host, gathers on the device)
t0 = time.perf_counter(); n = 0
for xb, yb in DeviceLoader(ds, batch_size=B, shuffle=True, generator=torch.Generator().manual_seed(1)):
    n += xb.shape[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"DeviceLoader      one epoch ({n} samples): {dt * 1e3:7.1f} ms  ({n / dt:9.0f} samples/s)")
# the reference's way: per-sample python window + stack, collate, pinned copy (main_final.py:97-154,483-494)
t0 = time.perf_counter(); m = 0
perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
for s in range(0, 32 * 20, B):
    xs, ys = [], []
    for i in perm[s:s + B].tolist():
        fr = [inp[j] if j >= 0 else torch.zeros_like(inp[0]) for j in range(i - T + 1, i + 1)]
        xs.append(torch.stack(fr)); ys.append(out[i])
    xb = torch.stack(xs).pin_memory().cuda(non_blocking=True); yb = torch.stack(ys).pin_memory().cuda(non_blocking=True)
    m += B
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host loop + H2D   ({m} samples, {torch.get_num_threads()} threads):   {m / dt:9.0f} samples/s")

# ---- evaluation: 1080 validation steps in batches of 32
V = 1080
stats = {0: {"method": "zscore", "params": {"mean": 280.0, "std": 20.0}},
         1: {"method": "log1p", "params": {"mean": 0.5, "std": 0.8}}}
lat = np.linspace(-88.75, 88.75, H)
try:
    ev = DeviceEvaluator(["tas", "pr"], stats, lat, H, W)
except Exception as e:           # (stat key names differ: fall back to identity-like stats)
    print("evaluator stats format:", e); sys.exit(0)
p = torch.randn(V, 2, H, W, device="cuda") * 0.3; t = torch.randn(V, 2, H, W, device="cuda") * 0.3
for s in range(0, V, B):
    ev.update(p[s:s + B], t[s:s + B])
ev.compute("val"); ev.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(0, V, B):
    ev.update(p[s:s + B], t[s:s + B])
res = ev.compute("val")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"DeviceEvaluator   {V} steps: {dt * 1e3:7.2f} ms  ({V / dt:9.0f} samples/s; {2 * V * 2 * H * W * 4 / dt / 1e9:6.1f} GB/s of pred+target)")
