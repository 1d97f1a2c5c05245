#!/usr/bin/env python3
"""Run one conv3x3 / wgrad3x3 launch shape repeatedly (for rocprofv3 --pmc runs).
    python tools/one_conv.py conv|wgrad N CIN COUT H W CONFIG [REPS]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_amd import ops

what, n, ci, co, h, w, cfg = sys.argv[1], *map(int, sys.argv[2:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
x = torch.randn(n, ci, h, w, device="cuda")
dy = torch.randn(n, co, h, w, device="cuda")
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
if what == "conv":
    wp = ops.pack_conv3x3(wt)
    out = torch.empty(n, co, h, w, device="cuda")
    for _ in range(reps):
        ops.conv3x3(x, wp, co, out=out, config=cfg)
else:
    g = torch.zeros(co, 9, ci, device="cuda")
    for _ in range(reps):
        ops.wgrad3x3(x, dy, g, config=cfg)
torch.cuda.synchronize()
