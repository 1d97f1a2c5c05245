#!/usr/bin/env python3
"""EXPERIMENT: what a small-N conv launch costs as a function of the reduction split (lstm.h / decoder shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_amd import ops
from conv_microbench import timeit

for name, n, ci, co, h, w, cfgs in (("lstm.h", 16, 128, 512, 6, 9, (5, 27)), ("up3.c2", 16, 128, 128, 12, 18, (27, 3)),
                                    ("up2.c2", 16, 64, 64, 24, 36, (27, 0))):
    x = torch.randn(n, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    out = torch.zeros(n, co, h, w, device="cuda")
    wph, winv = ops.pack_conv3x3_h3(wt)
    for cfg in cfgs:
        row = f"{name} cfg {cfg:2d}:"
        for ks in (1, 2, 4, 8):
            c = ops.H3_BASE + cfg + (ks << 8)
            f = lambda: ops.conv3x3(x, None, co, out=out, wph=wph, winv=winv, config=c, out_zeroed=True)
            try:
                f(); t = timeit(f, 50)
                row += f"  ks{ks} {t:6.1f}"
            except Exception as e:
                row += f"  ks{ks} n/a"
        print(row, flush=True)
