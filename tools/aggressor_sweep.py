"""Which launches of the library disturb a co-resident workgroup?  Detector = tools/canary (a workgroup that parks known values
in 56 + VGPRs / re-checks fixed arithmetic), recorded into one hipGraph beside the candidate on a forked stream and replayed
REPLAYS times (tools/coresidency_probe.py has the pairing that led here: cm_block_tail_bwd beside the ConvLSTM's h-part
weight gradient).

    python tools/aggressor_sweep.py [--out gpurun_out/aggressor_sweep.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402
import coresidency_probe as cp  # noqa: E402

cp.REPLAYS = 20


def candidates():
    torch.manual_seed(1)
    out = {}
    n = 96
    # weight gradients, every configuration of the three families, at the H/8 and H/4 levels
    for (c0, cout, h, w) in ((128, 512, 6, 9), (256, 256, 6, 9), (128, 128, 12, 18), (64, 64, 24, 36)):
        x = torch.tanh(torch.randn(n, c0, h, w, device="cuda"))
        dy = torch.randn(n, cout, h, w, device="cuda")
        g = torch.zeros(cout, 9, c0, device="cuda")
        bex, bey = ops.SampleExponents.measure(x), ops.SampleExponents.measure(dy)
        for c in range(14):
            out[f"wgrad3x3 fp16x3 cfg {c:2d} {c0}->{cout} @{h}x{w}"] = (
                lambda x=x, dy=dy, g=g, bex=bex, bey=bey, c=c: ops.wgrad3x3(x, dy, g, be_x=bex, be_y=bey, config=ops.H3_BASE + c + (4 << 8)))
        if (h, w) == (6, 9) and c0 == 128:
            for c in range(14):
                out[f"wgrad3x3 bf16x6 cfg {c:2d} {c0}->{cout} @{h}x{w}"] = (
                    lambda x=x, dy=dy, g=g, c=c: ops.wgrad3x3(x, dy, g, config=ops.SPLIT_BASE + c + (4 << 8)))
            for c in range(ops.lib.cm_wgrad3x3_num_configs()):
                out[f"wgrad3x3 fp32 cfg {c:2d} {c0}->{cout} @{h}x{w}"] = (
                    lambda x=x, dy=dy, g=g, c=c: ops.wgrad3x3(x, dy, g, config=c + (4 << 8)))
    # forward / data-gradient convs, fp16x3 family
    for (c0, cout, h, w) in ((256, 256, 6, 9), (128, 128, 12, 18), (64, 64, 24, 36), (32, 32, 48, 72)):
        x = torch.randn(n, c0, h, w, device="cuda")
        wt = torch.randn(cout, c0, 3, 3, device="cuda") * 0.02
        wph, winv = ops.pack_conv3x3_h3(wt)
        for c in range(ops.lib.cm_conv3x3_split_num_configs()):
            for k in (0, 2):
                out[f"conv3x3 fp16x3 cfg {c:2d} ksplit {k} {c0}->{cout} @{h}x{w}"] = (
                    lambda x=x, wph=wph, winv=winv, cout=cout, c=c, k=k: ops.conv3x3(x, None, cout, wph=wph, winv=winv,
                                                                                      config=ops.H3_BASE + c + (k << 8)))
    # ConvLSTM fused steps
    b, ch = 16, 128
    wl = torch.randn(4 * ch, 3 * ch, 3, 3, device="cuda") * 0.02
    wph, winv = ops.pack_conv3x3_h3(wl, c_off=2 * ch, cin=ch)
    hp = torch.tanh(torch.randn(b, ch, 6, 9, device="cuda")); cprev = torch.randn(b, ch, 6, 9, device="cuda")
    gx = torch.randn(b, 4 * ch, 6, 9, device="cuda"); co = torch.empty_like(cprev); ho = torch.empty_like(cprev)
    out["lstm_step_fwd B=16"] = lambda: ops.lstm_step_fwd(hp, wph, winv, gx, cprev, co, ho)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "aggressor_sweep.txt"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        def say(*a):
            line = " ".join(str(v) for v in a)
            if "mismatch 0 of" not in line:
                print(line)
            f.write(line + "\n")
            f.flush()
        detectors = {"parked-register canary (125 VGPRs, 256 threads)": cp.canary_case(56, 256, 7700),
                     "arithmetic canary (v_pk_mul swizzle, 256 threads)": cp.pk_canary_case(0, threads=256),
                     "arithmetic canary (v_mul, 256 threads)": cp.pk_canary_case(6, threads=256)}
        for name, a in candidates().items():
            if args.only and args.only not in name:
                continue
            try:
                a()
                torch.cuda.synchronize()
            except RuntimeError as e:
                say(f"{name}: not applicable ({str(e)[:50]})")
                continue
            for dn, d in detectors.items():
                cp.paired_canary(d, a, say, f"{name} | {dn}")
