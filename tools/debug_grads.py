#!/usr/bin/env python3
"""Compare the gradient entering each encoder block (d s_k) between the HIP engine and the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import oracle
from climate_amd import engine
from climate_amd.model import AttUNetConvLSTM

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()

base, T, B, H, W = map(int, sys.argv[1:6])
P = oracle.closed_form_params(5, 2, base, salt=9)
gen = torch.Generator("cpu").manual_seed(321)
x = torch.randn(B, T, 5, H, W, generator=gen); y = torch.randn(B, 2, H, W, generator=gen)
p64 = {k: v.double().clone().requires_grad_() for k, v in P.items()}
out, inter = oracle.model_forward(p64, x.double(), return_intermediates=True)
for k in ("s1", "s2", "s3", "s4"):
    for t in inter[k]: t.retain_grad()
for k in ("d3", "d2", "d1"): inter[k].retain_grad()
F.mse_loss(out, y.double()).backward()
ref = {}
for k in ("s1", "s2", "s3", "s4"):
    ref[k] = torch.stack([t.grad for t in inter[k]], 1).reshape(B * T, *inter[k][0].shape[1:])   # [B*T, ...] n = b*T+t
captured = {}
orig = engine._block_bwd
def spy(p, pk, g, gw, ss, prefix, ctx, dout, need_dx=True):
    captured[prefix] = dout.clone()
    r = orig(p, pk, g, gw, ss, prefix, ctx, dout, need_dx)
    if r is not None: captured[prefix + "/dx"] = r.clone()
    return r
engine._block_bwd = spy
m = AttUNetConvLSTM(5, 2, base, T); m.load_state_dict(P); m = m.cuda()
F.mse_loss(m(x.cuda()), y.cuda()).backward()
for k, pre in (("s4", "enc4.conv."), ("s3", "enc3.conv."), ("s2", "enc2.conv."), ("s1", "enc1.")):
    print(f"d{k}: rel err {rel(captured[pre], ref[k]):.2e}")
for k, pre in (("d1", "up1.conv."), ("d2", "up2.conv."), ("d3", "up3.conv.")):
    print(f"d{k}: rel err {rel(captured[pre], inter[k].grad):.2e}")
named = dict(m.named_parameters())
for k in ("enc1.body.0.weight", "enc1.body.4.bias", "enc2.conv.body.0.weight", "up1.conv.body.0.weight"):
    print(k, f"{rel(named[k].grad, p64[k].grad):.2e}")
