#!/usr/bin/env python3
"""EXPERIMENT: tuned conv3x3 (atomics for reduction splits) against the partial-slices form at the half-batch shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_amd import ops
from conv_microbench import layers, timeit

for name, n, c0, c1, co, h, w in layers(32, 16, 6):
    if c1 or c0 < 32:
        continue
    for tag, (ci_, co_) in (("fwd", (c0, co)), ("dgrad", (co, c0))):
        x = torch.randn(n, ci_, h, w, device="cuda")
        wt = torch.randn(co_, ci_, 3, 3, device="cuda") * 0.05
        wph, winv = ops.pack_conv3x3_h3(wt)
        out = torch.empty(n, co_, h, w, device="cuda")
        f = lambda: ops.conv3x3(x, None, co_, out=out, wph=wph, winv=winv)
        f(); t0 = timeit(f, 20)
        r = ops.conv3x3_parts(x, co_, wph, winv)
        if r is None:
            print(f"{name:8s} {tag:5s} N{n:3d} {ci_:4d}->{co_:4d} {h}x{w}: tuned {t0:6.1f} us  parts n/a"); continue
        parts = r[0]
        g = lambda: ops.conv3x3_parts(x, co_, wph, winv, parts=parts)
        g(); t1 = timeit(g, 20)
        key = ("conv3x3p", n, h, w, ci_, 0, co_)
        print(f"{name:8s} {tag:5s} N{n:3d} {ci_:4d}->{co_:4d} {h}x{w}: tuned {t0:6.1f} us (cfg {ops.LAST_CONV_CONFIG - ops.H3_BASE:5d})  "
              f"parts {t1:6.1f} us (k {r[1]}, cfg {(ops._TUNED[key] - ops.H3_BASE) & 255})", flush=True)
