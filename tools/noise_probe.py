"""Where does the device path's rounding noise enter?  Forward intermediates of the hot-path model at a realistic size
(device vs the float64 oracle, beside the float32 CPU oracle vs float64), then the gradients in network order.

    python tools/noise_probe.py [--base 64] [--T 6] [--H 192] [--W 288]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import oracle  # noqa: E402
from oracle.cpu_ref import _gn_silu  # noqa: E402
from climate_amd import engine, ops  # noqa: E402
from climate_amd.model import AttUNetConvLSTM  # noqa: E402
from _decisions import hip_decisions  # noqa: E402


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", type=int, default=64); ap.add_argument("--T", type=int, default=6)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--H", type=int, default=192); ap.add_argument("--W", type=int, default=288)
    ap.add_argument("--salt", type=int, default=9)
    a = ap.parse_args()
    gen = torch.Generator("cpu").manual_seed(321)
    x = torch.randn(a.B, a.T, 5, a.H, a.W, generator=gen); y = torch.randn(a.B, 2, a.H, a.W, generator=gen)
    P = oracle.closed_form_params(5, 2, a.base, salt=a.salt)
    m = AttUNetConvLSTM(5, 2, a.base, a.T)
    m.load_state_dict(P)
    m = m.cuda()
    p = m._param_dict()
    g = m._views(torch.zeros(m.n_flat_trainable, device="cuda"))
    pk = engine.get_plan(p, None, False).pack()
    B, T = a.B, a.T
    xd = x.cuda().view(B * T, 5, a.H, a.W)
    s1, c1, p1 = engine._block_fwd(p, pk, "enc1.", xd, None, True, pool=True)
    s2, c2, p2 = engine._block_fwd(p, pk, "enc2.conv.", p1, None, True, pool=True)
    s3, c3, p3 = engine._block_fwd(p, pk, "enc3.conv.", p2, None, True, pool=True)
    s4, c4 = engine._block_fwd(p, pk, "enc4.conv.", p3, None, True)
    k1, k2, k3 = (ops.time_mean(sk, B, T) for sk in (s1, s2, s3))
    bott, _ = engine.convlstm_fwd(p, pk, s4, B, T, True)
    d3, _ = engine.up_fwd(p, pk, "up3.", bott, k3, True)
    d2, _ = engine.up_fwd(p, pk, "up2.", d3, k2, True)
    d1, _ = engine.up_fwd(p, pk, "up1.", d2, k1, True)
    pred = ops.head_fwd(d1, p["head.weight"], p["head.bias"])
    dev = dict(s1=s1, s2=s2, s3=s3, s4=s4, bott=bott, d3=d3, d2=d2, d1=d1, pred=pred)
    with torch.no_grad():
        o64, i64 = oracle.model_forward({k: v.double() for k, v in P.items()}, x.double(), return_intermediates=True)
        o32, i32 = oracle.model_forward(P, x, return_intermediates=True)

    def stack(i, o):
        return dict(s1=torch.stack(i["s1"], 1).flatten(0, 1), s2=torch.stack(i["s2"], 1).flatten(0, 1),
                    s3=torch.stack(i["s3"], 1).flatten(0, 1), s4=torch.stack(i["s4"], 1).flatten(0, 1),
                    bott=i["lstm_out"][-1], d3=i["d3"], d2=i["d2"], d1=i["d1"], pred=o)
    r64, r32 = stack(i64, o64), stack(i32, o32)
    print(f"numerics {engine.NUMERICS}; forward rel-L2 vs float64:        device     cpu-fp32")
    for k in dev:
        print(f"   {k:6s} {rel(dev[k], r64[k]):.2e}   {rel(r32[k], r64[k]):.2e}")
    # one Up block on EXACT inputs (the oracle's float64 activations rounded to float32), stage by stage
    def stages(pp, xin, skip, prefix, dt):
        pr = prefix + "conv."
        q = {k: v.to(dt) for k, v in pp.items() if k.startswith(prefix)}
        u = F.conv_transpose2d(xin.to(dt), q[prefix + "up.weight"], q[prefix + "up.bias"], stride=2)
        y1 = F.conv2d(torch.cat([u, skip.to(dt)], 1), q[pr + "body.0.weight"], padding=1)
        a1 = _gn_silu(y1, q[pr + "body.1.weight"], q[pr + "body.1.bias"])
        y2 = F.conv2d(a1, q[pr + "body.3.weight"], padding=1)
        a2 = _gn_silu(y2, q[pr + "body.4.weight"], q[pr + "body.4.bias"])
        z = oracle.se_block(a2, q[pr + "se.fc.0.weight"], q[pr + "se.fc.2.weight"])
        out = oracle.spatial_gate(z, q[pr + "spat.conv.weight"])
        return dict(u=u, y1=y1, a1=a1, y2=y2, a2=a2, out=out)
    for prefix, xin, skip in (("up3.", r64["bott"], torch.stack(i64["s3"], 0).mean(0)),
                              ("up2.", r64["d3"], torch.stack(i64["s2"], 0).mean(0)),
                              ("up1.", r64["d2"], torch.stack(i64["s1"], 0).mean(0))):
        xin32, skip32 = xin.float(), skip.float()
        with torch.no_grad():
            e64 = stages(P, xin32, skip32, prefix, torch.float64)
            e32 = stages(P, xin32, skip32, prefix, torch.float32)
        outd, (ctx, _) = engine.up_fwd(p, pk, prefix, xin32.cuda(), skip32.cuda(), True)
        devs = dict(u=ctx.x0, y1=ctx.y1, a1=ctx.a1, y2=ctx.y2, a2=ctx.activation2(), out=outd)
        print(f"{prefix} on exact inputs, rel-L2 vs float64:   device     cpu-fp32    (y: also relative to the centred value)")
        for k in devs:
            extra = ""
            if k in ("y1", "y2"):
                c = (e64[k] - e64[k].mean((1, 2, 3), keepdim=True))
                extra = f"   std/rms {(c.norm() / e64[k].norm()).item():.3f}"
            print(f"   {k:4s} {rel(devs[k], e64[k]):.2e}   {rel(e32[k], e64[k]):.2e}{extra}")
    # gradients
    pred2, sv = engine.forward(p, pk, x.cuda(), save=True)
    dec = hip_decisions(sv)
    yd = y.cuda()
    engine.backward(p, pk, g, sv, (2.0 / pred2.numel()) * (pred2 - yd))
    torch.cuda.synchronize()
    pc = {k: v.double().requires_grad_() for k, v in P.items()}
    oracle.training_loss(pc, x.double(), y.double(), decisions=dec).backward()
    p32 = {k: v.float().requires_grad_() for k, v in P.items()}
    oracle.training_loss(p32, x, y, decisions=hip_decisions(sv)).backward()
    print("gradient rel-L2 vs float64 (network order):     device     cpu-fp32")
    for k in pc:
        if pc[k].grad is not None:
            print(f"   {k:34s} {rel(g[k], pc[k].grad):.2e}   {rel(p32[k].grad, pc[k].grad):.2e}")


if __name__ == "__main__":
    main()
