"""Ablation timing of cm_conv3x3_split (CM_CONVS_DBG bits: 1 skip loads, 2 skip MFMA, 8 skip convert+store)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd import ops
cases = {"enc3.c2": (192, 128, 128, 12, 18, [3, 18]), "enc1.c2": (192, 32, 32, 48, 72, [9, 3]), "enc4.c2": (192, 256, 256, 6, 9, [16, 7]),
         "lstm.x": (192, 256, 512, 6, 9, [7, 5])}
for name, (n, ci, co, h, w, cfgs) in cases.items():
    x = torch.randn(n, ci, h, w, device="cuda"); wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wps = ops.pack_conv3x3_split(wt); out = torch.empty(n, co, h, w, device="cuda")
    for cfg in cfgs:
        f = lambda: ops.conv3x3_split(x, wps, co, out=out, config=cfg)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"dbg={os.environ.get('CM_CONVS_DBG', '0'):>2s} {name} cfg {cfg:2d}: {e0.elapsed_time(e1) * 50:.1f} us")
