#!/usr/bin/env python3
"""Per layer of BASELINE config 2: the tuned conv / data-gradient / weight-gradient launch against its two bounds
(HBM: compulsory bytes at 6 TB/s; MFMA: 3 (fp16x3) or 6 (bf16x6) products at 2.5 PFLOP/s).

    python tools/layer_bounds.py [--base 32] [--B 32] [--T 6]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402
from conv_microbench import layers, timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--base", type=int, default=32); ap.add_argument("--B", type=int, default=32)
ap.add_argument("--T", type=int, default=6)
a = ap.parse_args()
tot = dict(conv=0.0, dgrad=0.0, wgrad=0.0, bconv=0.0, bdgrad=0.0, bwgrad=0.0)
wprod = 3 if ops.WGRAD_H3 else 6
print(f"{'layer':9s} {'shape':24s} | conv us (hbm, mfma bound) | dgrad us | wgrad us (x{wprod})")
for name, n, c0, c1, co, h, w in layers(a.base, a.B, a.T):
    ci = c0 + c1
    x0 = torch.randn(n, c0, h, w, device="cuda")
    x1 = torch.randn(n, c1, h, w, device="cuda") if c1 else None
    dy = torch.randn(n, co, h, w, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    flops = 2.0 * n * h * w * co * ci * 9
    byts = 4.0 * n * h * w * (ci + co)
    hb = byts / 6e12 * 1e6
    row = f"{name:9s} N{n:3d} {ci:4d}->{co:4d} {h:2d}x{w:2d} |"
    small = ci * 9 <= 64
    # forward
    wph, winv = ops.pack_conv3x3_h3(wt)
    out = torch.empty(n, co, h, w, device="cuda")
    f = lambda: ops.conv3x3(x0, None, co, x1=x1, out=out, wph=wph, winv=winv, w_raw=wt if small else None)
    f(); t = timeit(f, 10); tot["conv"] += t
    mb = 3 * flops / 2.5e15 * 1e6
    tot["bconv"] += max(hb, mb)
    row += f" {t:6.1f} ({hb:5.1f}, {mb:5.1f}) x{t / max(hb, mb):4.1f} |"
    # data gradient (not for the first layer)
    if name != "enc1.c1":
        wd, wdinv = ops.pack_conv3x3_h3(wt, dgrad=True)
        dx = torch.empty(n, ci, h, w, device="cuda")
        f = lambda: ops.conv3x3(dy, None, ci, out=dx, wph=wd, winv=wdinv)
        f(); t = timeit(f, 10); tot["dgrad"] += t; tot["bdgrad"] += max(hb, mb)
        row += f" {t:6.1f} x{t / max(hb, mb):4.1f} |"
    else:
        row += "               |"
    g = torch.zeros(co, 9, ci, device="cuda")
    bx = by = None
    if ops.WGRAD_H3:        # (in the engine the fp16x3 convs publish these while reading the same tensors)
        bx = ops.SampleExponents.measure(torch.cat([x0, x1], 1) if x1 is not None else x0)
        by = ops.SampleExponents.measure(dy)
    f = lambda: ops.wgrad3x3(x0, dy, g, x1=x1, be_x=bx, be_y=by)
    f(); t = timeit(f, 10); tot["wgrad"] += t
    mbw = wprod * flops / 2.5e15 * 1e6
    tot["bwgrad"] += max(hb, mbw)
    row += f" {t:6.1f} ({hb:5.1f}, {mbw:5.1f}) x{t / max(hb, mbw):4.1f}  cfg {ops.LAST_CONV_CONFIG}"
    print(row, flush=True)
print(f"totals: conv {tot['conv']:.0f} us (bound {tot['bconv']:.0f}), dgrad {tot['dgrad']:.0f} (bound {tot['bdgrad']:.0f}), "
      f"wgrad {tot['wgrad']:.0f} (bound {tot['bwgrad']:.0f});  lstm.h rows run T={a.T} times per step")
