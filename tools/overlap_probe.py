"""Probe for the round-1 side-stream failure (eager, CM_OVERLAP_WGRAD schedule, bf16x6 weight gradients co-resident
with the main chain): does the gate backward ever see operands that differ from what the forward wrote?

The backward now derives its own channel maximum `umax` from a2*s (attention_gates.hip); the forward stored its
maximum in fmap[:, 1].  Identical operands => bit-identical maxima.  This probe counts, per step and per ConvBlock,
the pixels where the two differ, WITHOUT synchronising inside the step (device-side counters), and with --snap also
keeps forward-time clones of (a2, s, fmap) to tell which operand changed in memory.

    python tools/overlap_probe.py [--steps 6] [--snap] [--serial]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd import engine, ops  # noqa: E402
from climate_amd.config import synthetic_config  # noqa: E402
from climate_amd.model import get_model  # noqa: E402
from climate_amd.trainer import HotPathTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--snap", action="store_true")
ap.add_argument("--serial", action="store_true")
args = ap.parse_args()
engine.OVERLAP_WGRAD = not args.serial

cfg = synthetic_config(base_channels=32, seq_len=6)
gen = torch.Generator("cpu").manual_seed(7)
x = torch.randn(32, 6, 5, 48, 72, generator=gen).cuda()
y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()
torch.manual_seed(cfg.seed)
m = get_model(cfg).cuda()
tr = HotPathTrainer(m, lr=1e-3, use_graph=False)

counters = []          # (step, block index, device tensor [4]: umax mismatches, a2 changed, s changed, fmap changed)
snaps = {}
step_no = [0]
blk_no = [0]

orig_fwd, orig_bwd = ops.se_spatial_gate_fwd, ops.gates_bwd


def fwd(a2, pooled, w1, w2, w7, pool_out=False):
    res = orig_fwd(a2, pooled, w1, w2, w7, pool_out=pool_out)
    if args.snap:
        out, z, s, fmap, gate = res[:5]
        snaps[fmap.data_ptr()] = (a2.clone(), s.clone(), fmap.clone())
    return res


def bwd(dout, a2, s, z, pooled, gate, fmap, *rest, **kw):
    res = orig_bwd(dout, a2, s, z, pooled, gate, fmap, *rest, **kw)
    umax = res[1][0]
    c = torch.zeros(4, device="cuda")
    c[0] = (umax != fmap[:, 1]).sum()
    if args.snap:
        a2c, sc, fc = snaps.pop(fmap.data_ptr())
        c[1] = (a2 != a2c).sum()
        c[2] = (s != sc).sum()
        c[3] = (fmap != fc).sum()
    counters.append((step_no[0], blk_no[0], c))
    blk_no[0] += 1
    return res


ops.se_spatial_gate_fwd, ops.gates_bwd = fwd, bwd
for it in range(args.steps):
    step_no[0], blk_no[0] = it, 0
    loss = tr.step(x, y)
torch.cuda.synchronize()
bad = 0
for st, blk, c in counters:
    v = c.tolist()
    if any(v):
        bad += 1
        print(f"step {st} block {blk}: umax != forward max at {int(v[0])} pixels; elements changed since the forward: "
              f"a2 {int(v[1])}, s {int(v[2])}, fmap {int(v[3])}")
print(f"overlap={'off' if args.serial else 'on'} snap={args.snap}: {len(counters)} gate backward launches over "
      f"{args.steps} steps, {bad} with a mismatch; final loss {loss.item():.6f}; grads finite: "
      f"{bool(torch.isfinite(tr.grad).all())}")
