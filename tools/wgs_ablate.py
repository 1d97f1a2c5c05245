"""Ablation timing of cm_wgrad3x3_split (run with CM_WGS_DBG=0,1,2,4,8 and combinations): one layer, chosen configs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd import ops
cases = {"enc4.c2": (192, 256, 256, 6, 9, [0, 14, 15]), "enc3.c2": (192, 128, 128, 12, 18, [12, 14, 15])}
for name, (n, ci, co, h, w, cfgs) in cases.items():
    x = torch.randn(n, ci, h, w, device="cuda"); dy = torch.randn(n, co, h, w, device="cuda")
    g = torch.zeros(co, 9, ci, device="cuda")
    for cfg in cfgs:
        f = lambda: ops.wgrad3x3(x, dy, g, config=ops.SPLIT_BASE + cfg + (4 << 8))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"dbg={os.environ.get('CM_WGS_DBG', '0'):>2s} {name} cfg {cfg:2d}: {e0.elapsed_time(e1) * 50:.1f} us")
