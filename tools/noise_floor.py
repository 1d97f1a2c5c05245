#!/usr/bin/env python3
"""fp32 noise floor check: HIP gradients and the fp32 CPU oracle's gradients, both against the fp64 CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import oracle
from climate_amd.model import AttUNetConvLSTM

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()

base, T, B, H, W = map(int, sys.argv[1:6])
P = oracle.closed_form_params(5, 2, base, salt=9)
gen = torch.Generator("cpu").manual_seed(321)
x = torch.randn(B, T, 5, H, W, generator=gen); y = torch.randn(B, 2, H, W, generator=gen)
p32 = {k: v.clone().requires_grad_() for k, v in P.items()}
oracle.training_loss(p32, x, y).backward()
p64 = {k: v.double().clone().requires_grad_() for k, v in P.items()}
oracle.training_loss(p64, x.double(), y.double()).backward()
m = AttUNetConvLSTM(5, 2, base, T); m.load_state_dict(P); m = m.cuda()
F.mse_loss(m(x.cuda()), y.cuda()).backward()
named = dict(m.named_parameters())
rows = []
for k in p64:
    if p64[k].grad is None: continue
    rows.append((rel(named[k].grad, p64[k].grad), rel(p32[k].grad, p64[k].grad), rel(named[k].grad, p32[k].grad), k))
rows.sort(reverse=True)
print(f"{'hip vs f64':>11} {'cpu32 vs f64':>12} {'hip vs cpu32':>12}  tensor")
for r in rows[:12]:
    print(f"{r[0]:11.2e} {r[1]:12.2e} {r[2]:12.2e}  {r[3]}")
