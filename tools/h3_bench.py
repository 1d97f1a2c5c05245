#!/usr/bin/env python3
"""EXPERIMENT: bf16x6 vs fp16x3 numerics in the same conv kernel structure -- time (best tile config) and accuracy
(vs fp64 on two samples) per BASELINE config-2 layer.   python tools/h3_bench.py [--only enc3.c2,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from climate_amd import ops  # noqa: E402
from climate_amd._lib import check, lib  # noqa: E402
from conv_microbench import layers, timeit  # noqa: E402


def conv_h3(x0, wph, winv, co, x1, out, cfg):
    n, c0, h, w = x0.shape
    return lib.cm_conv3x3_h3(x0.data_ptr(), x0.stride(0), c0, None if x1 is None else x1.data_ptr(),
                             0 if x1 is None else x1.stride(0), 0 if x1 is None else x1.shape[1], wph.data_ptr(),
                             winv.data_ptr(), None, None, 0, out.data_ptr(), out.stride(0), None, 0, n, h, w, co, cfg,
                             torch.cuda.current_stream().cuda_stream)


ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--base", type=int, default=32)
args = ap.parse_args()
only = set(filter(None, args.only.split(",")))
tot = [0.0, 0.0]
cases = []
for name, n, c0, c1, co, h, w in layers(args.base, 32, 6):
    if name == "enc1.c1" or (only and name not in only):
        continue
    cases.append((name, n, c0, c1, co, h, w))
    cases.append((name + "/d", n, co, 0, c0 + c1, h, w))            # its data gradient: cout -> cin
for name, n, c0, c1, co, h, w in cases:
    ci = c0 + c1
    x0 = torch.randn(n, c0, h, w, device="cuda")
    x1 = torch.randn(n, c1, h, w, device="cuda") if c1 else None
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    out = torch.empty(n, co, h, w, device="cuda")
    flops = 2.0 * n * h * w * co * ci * 9
    wps3 = ops.pack_conv3x3_split(wt)
    wph, winv = ops.pack_conv3x3_h3(wt)
    xc = torch.cat([x0, x1], 1) if c1 else x0
    ref = F.conv2d(xc[:2].double().cpu(), wt.double().cpu(), padding=1)
    res3, res2 = [], []
    for cfg in range(lib.cm_conv3x3_split_num_configs()):
        try:
            res3.append((timeit(lambda: ops.conv3x3_split(x0, wps3, co, x1=x1, out=out, config=cfg)), cfg))
        except RuntimeError:
            pass
        if conv_h3(x0, wph, winv, co, x1, out, cfg) == 0:
            res2.append((timeit(lambda: conv_h3(x0, wph, winv, co, x1, out, cfg)), cfg))
    res3.sort(); res2.sort()
    ops.conv3x3_split(x0, wps3, co, x1=x1, out=out, config=res3[0][1])
    e3 = ((out[:2].double().cpu() - ref).norm() / ref.norm()).item()
    conv_h3(x0, wph, winv, co, x1, out, res2[0][1])
    e2 = ((out[:2].double().cpu() - ref).norm() / ref.norm()).item()
    tot[0] += res3[0][0]; tot[1] += res2[0][0]
    print(f"{name:10s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: bf16x6 cfg {res3[0][1]:2d} {res3[0][0]:7.1f} us {flops / res3[0][0] / 1e6:6.1f} TF "
          f"err {e3:.1e} | fp16x3 cfg {res2[0][1]:2d} {res2[0][0]:7.1f} us {flops / res2[0][0] / 1e6:6.1f} TF err {e2:.1e}"
          f" | next: " + " ".join(f"{c}:{t:.0f}" for t, c in res2[1:4]))
print(f"sum of best: bf16x6 {tot[0]:.1f} us, fp16x3 {tot[1]:.1f} us")
