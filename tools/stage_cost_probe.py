#!/usr/bin/env python3
"""EXPERIMENT: fixed cost vs cost per 16-channel stage of a small fp16x3 conv launch, timed inside a replayed hipGraph
(50 launches back to back: device time, no host gaps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_amd import ops


def graph_time(fn, reps=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3


for (n, h, w, co, cfg) in ((16, 24, 36, 64, 27), (16, 6, 9, 512, 5), (96, 6, 9, 256, 5), (16, 48, 72, 32, 0)):
    row = f"N{n:3d} {h}x{w} cout {co:3d} cfg {cfg:2d}:"
    for ci in (16, 32, 64, 128, 256):
        x = torch.randn(n, ci, h, w, device="cuda")
        wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        out = torch.empty(n, co, h, w, device="cuda")
        wph, winv = ops.pack_conv3x3_h3(wt)
        f = lambda: ops.conv3x3(x, None, co, out=out, wph=wph, winv=winv, config=ops.H3_BASE + cfg)
        row += f"  cin{ci:3d} {graph_time(f):6.1f}"
    print(row + "  us per launch", flush=True)
# an empty-ish kernel for the launch floor inside a graph
z = torch.empty(1024, device="cuda")
from climate_amd._lib import lib
print(f"cm_zero of 4 KB inside the graph: {graph_time(lambda: lib.cm_zero(z.data_ptr(), 4096, torch.cuda.current_stream().cuda_stream)):.1f} us per launch")
