"""Soak run of the fused trainer: many graph-replayed steps on a fixed synthetic regression task; checks that the loss
stays finite, goes down, and that eager and graph modes agree step for step at the start."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd.config import synthetic_config
from climate_amd.model import get_model
from climate_amd.trainer import HotPathTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = synthetic_config(base_channels=32, seq_len=6)
torch.manual_seed(0)
gen = torch.Generator("cpu").manual_seed(7)
x = torch.randn(32, 6, 5, 48, 72, generator=gen).cuda()
y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()          # a learnable target
losses = {}
for mode in (True, False):
    torch.manual_seed(cfg.seed)
    m = get_model(cfg).cuda()
    tr = HotPathTrainer(m, lr=1e-3, use_graph=mode)
    ls = []
    for i in range(steps if mode else 6):
        ls.append(tr.step(x, y).item())
    losses[mode] = ls
g, e = losses[True], losses[False]
print("graph :", " ".join(f"{v:.5f}" for v in g[:6]), "...", " ".join(f"{v:.5f}" for v in g[-3:]))
print("eager :", " ".join(f"{v:.5f}" for v in e))
assert all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(g[:6], e)), "graph and eager diverge"
assert all(v == v and v < 1e6 for v in g), "non-finite loss"
assert steps < 200 or g[-1] < 0.7 * g[0], "loss did not go down"
print(f"soak ok: {steps} steps, loss {g[0]:.4f} -> {g[-1]:.4f}")
