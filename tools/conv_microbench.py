#!/usr/bin/env python3
"""Per-layer microbenchmark of the MFMA conv3x3 / wgrad3x3 launchers at BASELINE config-2 shapes (all tile configs).

    python tools/conv_microbench.py [--base 32] [--B 32] [--T 6] [--what conv|wgrad|both]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402
from climate_amd._lib import check, lib  # noqa: E402


def layers(b, B, T, H=48, W=72):
    N = B * T
    L = [("enc1.c1", N, 5, 0, b, H, W), ("enc1.c2", N, b, 0, b, H, W),
         ("enc2.c1", N, b, 0, 2 * b, H // 2, W // 2), ("enc2.c2", N, 2 * b, 0, 2 * b, H // 2, W // 2),
         ("enc3.c1", N, 2 * b, 0, 4 * b, H // 4, W // 4), ("enc3.c2", N, 4 * b, 0, 4 * b, H // 4, W // 4),
         ("enc4.c1", N, 4 * b, 0, 8 * b, H // 8, W // 8), ("enc4.c2", N, 8 * b, 0, 8 * b, H // 8, W // 8),
         ("lstm.x", N, 8 * b, 0, 16 * b, H // 8, W // 8), ("lstm.h", B, 4 * b, 0, 16 * b, H // 8, W // 8),
         ("up3.c1", B, 4 * b, 4 * b, 4 * b, H // 4, W // 4), ("up3.c2", B, 4 * b, 0, 4 * b, H // 4, W // 4),
         ("up2.c1", B, 2 * b, 2 * b, 2 * b, H // 2, W // 2), ("up2.c2", B, 2 * b, 0, 2 * b, H // 2, W // 2),
         ("up1.c1", B, b, b, b, H, W), ("up1.c2", B, b, 0, b, H, W)]
    return L


def timeit(fn, reps=5):
    fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", type=int, default=32)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=6)
    ap.add_argument("--what", default="both")
    ap.add_argument("--split", action="store_true", help="also time the bf16x6 kernel on the forward shapes")
    ap.add_argument("--all", action="store_true", help="print every configuration's time, not only the best six")
    ap.add_argument("--only", default="", help="comma-separated layer names (default: all)")
    ap.add_argument("--skip-fp32", action="store_true", help="with --split: time only the bf16x6 kernels")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    st = torch.cuda.current_stream().cuda_stream
    tot = {"conv": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, n, c0, c1, co, h, w in layers(args.base, args.B, args.T):
        if only and name not in only:
            continue
        ci = c0 + c1
        x0 = torch.randn(n, c0, h, w, device="cuda")
        x1 = torch.randn(n, c1, h, w, device="cuda") if c1 else None
        dy = torch.randn(n, co, h, w, device="cuda")
        wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        flops = 2.0 * n * h * w * co * ci * 9
        if args.what in ("conv", "both"):
            wp = ops.pack_conv3x3(wt)
            out = torch.empty(n, co, h, w, device="cuda")
            res = []
            for cfg in range(lib.cm_conv3x3_num_configs()):
                try:
                    t = timeit(lambda: ops.conv3x3(x0, wp, co, x1=x1, out=out, config=cfg))
                except RuntimeError:
                    continue
                res.append((t, cfg))
            res.sort()
            tot["conv"] += res[0][0]
            print(f"conv  {name:8s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: best cfg {res[0][1]:2d} {res[0][0]:7.1f} us "
                  f"{flops / res[0][0] / 1e6:6.1f} TF | " + " ".join(f"{c}:{t:.0f}" for t, c in res[1:6]))
            if x1 is None and ci * 9 <= 64:
                t = timeit(lambda: ops.conv3x3(x0, None, co, out=out, w_raw=wt, config=ops.SMALLC_CFG))
                print(f"convsc {name:8s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: {t:7.1f} us {flops / t / 1e6:6.1f} TF "
                      f"(cm_conv3x3_smallc)")
            if args.split:
                wps = ops.pack_conv3x3_split(wt)
                res = []
                for cfg in range(lib.cm_conv3x3_split_num_configs()):
                    try:
                        t = timeit(lambda: ops.conv3x3_split(x0, wps, co, x1=x1, out=out, config=cfg))
                    except RuntimeError:
                        continue
                    res.append((t, cfg))
                res.sort()
                tot.setdefault("conv_split", 0.0)
                tot["conv_split"] += res[0][0]
                print(f"split {name:8s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: best cfg {res[0][1]:2d} {res[0][0]:7.1f} us "
                      f"{flops / res[0][0] / 1e6:6.1f} TF | " + " ".join(f"{c}:{t:.0f}" for t, c in res[1:6]))
            if name != "enc1.c1":
                wpd = ops.pack_conv3x3(wt, dgrad=True)
                outd = torch.empty(n, ci, h, w, device="cuda")
                res = []
                for cfg in range(lib.cm_conv3x3_num_configs()):
                    t = timeit(lambda: ops.conv3x3(dy, wpd, ci, out=outd, config=cfg))
                    res.append((t, cfg))
                res.sort()
                tot["dgrad"] += res[0][0]
                print(f"dgrad {name:8s} N={n:3d} {co:3d}->{ci:3d} @{h}x{w}: best cfg {res[0][1]:2d} {res[0][0]:7.1f} us "
                      f"{flops / res[0][0] / 1e6:6.1f} TF | " + " ".join(f"{c}:{t:.0f}" for t, c in res[1:6]))
        if args.what in ("wgrad", "both"):
            g = torch.zeros(co, 9, ci, device="cuda")
            res = []
            for cfg in range(0 if args.skip_fp32 else lib.cm_wgrad3x3_num_configs()):
                for upb in (2, 3, 4, 6, 8):
                    t = timeit(lambda: ops.wgrad3x3(x0, dy, g, x1=x1, config=cfg + (upb << 8)))
                    res.append((t, f"{cfg}/{upb}"))
            res.sort()
            res = res or [(float("inf"), "-")]
            tot["wgrad"] += res[0][0] if res[0][1] != "-" else 0.0
            print(f"wgrad {name:8s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: best cfg {res[0][1]:>5s} {res[0][0]:7.1f} us "
                  f"{flops / res[0][0] / 1e6:6.1f} TF | " + " ".join(f"{c}:{t:.0f}" for t, c in res[1:5]))
            if x1 is None and ci * 9 <= 64 and w % 4 == 0:
                t = timeit(lambda: ops.wgrad3x3(x0, dy, g, config=ops.SMALLC_CFG))
                print(f"wgsc  {name:8s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: {t:7.1f} us {flops / t / 1e6:6.1f} TF "
                      f"(cm_wgrad3x3_smallc, two launches)")
            if args.split and (x1 is None or x0.shape[1] % 32 == 0):
                res = []
                for cfg in range(lib.cm_wgrad3x3_split_num_configs()):
                    for upb in (2, 4, 8):
                        t = timeit(lambda: ops.wgrad3x3(x0, dy, g, x1=x1, config=ops.SPLIT_BASE + cfg + (upb << 8)))
                        res.append((t, f"{cfg}/{upb}"))
                res.sort()
                tot.setdefault("wgrad_split", 0.0)
                tot["wgrad_split"] += res[0][0]
                print(f"wgs   {name:8s} N={n:3d} {ci:3d}->{co:3d} @{h}x{w}: best cfg {res[0][1]:>5s} {res[0][0]:7.1f} us "
                      f"{flops / res[0][0] / 1e6:6.1f} TF | " + " ".join(f"{c}:{t:.0f}" for t, c in res[1:(99 if args.all else 6)]))
    print("sum of best (us):", {k: round(v, 1) for k, v in tot.items()}, "(lstm.h counted once; it runs T-1 times)")


if __name__ == "__main__":
    main()
