#!/usr/bin/env python3
"""Run one conv3x3 / wgrad3x3 launch shape repeatedly (for rocprofv3 --pmc runs).
    python tools/one_conv.py conv|h3|split|wgrad|wgh3 N CIN COUT H W CONFIG [REPS]"""
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
out = torch.empty(n, co, h, w, device="cuda")
if what == "conv":
    wp = ops.pack_conv3x3(wt)
    for _ in range(reps):
        ops.conv3x3(x, wp, co, out=out, config=cfg)
elif what == "split":
    wps = ops.pack_conv3x3_split(wt)
    for _ in range(reps):
        ops.conv3x3(x, None, co, out=out, wps=wps, config=ops.SPLIT_BASE + cfg)
elif what == "h3":
    wph, winv = ops.pack_conv3x3_h3(wt)
    for _ in range(reps):
        ops.conv3x3(x, None, co, out=out, wph=wph, winv=winv, config=ops.H3_BASE + cfg)
else:
    g = torch.zeros(co, 9, ci, device="cuda")
    base = ops.H3_BASE if what == "wgh3" else (ops.SPLIT_BASE if what == "wgs" else 0)
    for _ in range(reps):
        ops.wgrad3x3(x, dy, g, config=base + cfg)
torch.cuda.synchronize()
