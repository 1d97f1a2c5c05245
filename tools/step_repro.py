"""Replay-to-replay reproducibility of the benchmarked step (BASELINE config 2, hipGraph, two micro-batches on two streams)
with the learning rate at ZERO: every replay computes the gradient of the same parameters on the same batch, so the replays
may differ by the order of float atomics only.  Reported per run: the largest per-tensor difference of any replay's gradient
from the first one's (relative to the step's gradient scale).  Anything well above ~1e-6 is a schedule-dependent result
(before round 3's fix: the packed-fp32 defect of profiles/r03/coresidency/ whenever the two halves' kernels met badly).

    python tools/step_repro.py [--steps 40] [--pairs 3] [--out gpurun_out/step_repro.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd.config import synthetic_config  # noqa: E402
from climate_amd.model import get_model  # noqa: E402
from climate_amd.trainer import HotPathTrainer  # noqa: E402


def run(cfg, x, y, steps):
    torch.manual_seed(cfg.seed)
    m = get_model(cfg).cuda()
    tr = HotPathTrainer(m, lr=0.0, use_graph=True, distributed=False, micro_batches=2)
    g0, worst, name, lmin, lmax = None, 0.0, "", 1e30, 0.0
    for i in range(steps):
        loss = tr.step(x, y).item()
        lmin, lmax = min(lmin, loss), max(lmax, loss)
        g = {k: v.clone() for k, v in m._views(tr.grad).items()}
        if g0 is None:
            g0 = g
            big = max(v.norm().item() for v in g0.values())
            continue
        for k in g0:
            e = (g[k] - g0[k]).norm().item() / max(g0[k].norm().item(), 1e-3 * big)
            if e > worst:
                worst, name = e, k
    return worst, name, (lmax - lmin) / lmin


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "step_repro.txt"))
    args = ap.parse_args()
    cfg = synthetic_config(base_channels=32, seq_len=6)
    gen = torch.Generator("cpu").manual_seed(7)
    x = torch.randn(32, 6, 5, 48, 72, generator=gen).cuda()
    y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        def say(*a):
            line = " ".join(str(v) for v in a)
            print(line)
            f.write(line + "\n")
            f.flush()
        say(f"library: {os.environ.get('CM_LIB_TAG', '') or 'shipped build'}")
        for p in range(args.pairs):
            worst, name, dl = run(cfg, x, y, args.steps)
            say(f"run {p}: {args.steps} replays, learning rate 0: worst gradient difference from the first replay {worst:.1e} "
                f"({name}); loss spread {dl:.1e}")
