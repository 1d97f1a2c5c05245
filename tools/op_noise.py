"""Per-launcher rounding noise at decoder-sized shapes: device vs float64, beside torch-CPU float32 vs float64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from climate_amd import ops  # noqa: E402


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def centred(a, b):
    """error relative to the CENTRED reference (what a following GroupNorm sees)."""
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).norm() / (b - b.mean()).norm()).item()


g = torch.Generator("cpu").manual_seed(0)
for (n, c0, c1, co, h, w) in ((6, 256, 256, 256, 48, 72), (2, 128, 128, 128, 96, 144), (6, 128, 0, 128, 96, 144),
                              (2, 1024, 0, 2048, 24, 36)):
    x = F.silu(torch.randn(n, c0 + c1, h, w, generator=g))
    wt = torch.randn(co, c0 + c1, 3, 3, generator=g) / (3 * (c0 + c1) ** 0.5)
    b = torch.randn(co, generator=g) * 0.1
    y64 = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    y32 = F.conv2d(x, wt, b, padding=1)
    print(f"conv3x3 N{n} C{c0}+{c1}->{co} {h}x{w}: cpu-fp32 {rel(y32, y64):.2e} (centred {centred(y32, y64):.2e})")
    xd = x.cuda(); wd = wt.cuda(); bd = b.cuda()
    x0 = xd[:, :c0]; x1 = xd[:, c0:] if c1 else None
    wp = ops.pack_conv3x3(wd)
    wps = ops.pack_conv3x3_split(wd)
    wph, winv = ops.pack_conv3x3_h3(wd)
    nk = (c0 + c1) // 16
    for name, base, ksplits in (("fp32", 0, (1, 4)), ("bf16x6", ops.SPLIT_BASE, (1, 4)), ("fp16x3", ops.H3_BASE, (1, 4, 16))):
        for ks in ksplits:
            if ks > 1 and nk < 2 * ks:
                continue
            cfg = base + ((ks if ks > 1 else 0) << 8)
            out = torch.zeros(n, co, h, w, device="cuda")
            y = ops.conv3x3(x0, wp, co, x1=x1, bias=bd, out=out, config=cfg, wps=wps, wph=wph, winv=winv, out_zeroed=True)
            print(f"     {name:7s} ksplit {ks:2d}: {rel(y, y64):.2e} (centred {centred(y, y64):.2e})")
    # GroupNorm + SiLU on that output
    gam = 1 + 0.1 * torch.randn(co, generator=g); bet = 0.1 * torch.randn(co, generator=g)
    z64 = F.silu(F.group_norm(y64, 8, gam.double(), bet.double()))
    z32 = F.silu(F.group_norm(y64.float(), 8, gam, bet))
    zd, _ = ops.gn_silu_fwd(y64.float().cuda(), gam.cuda(), bet.cuda())[:2]
    print(f"   gn_silu on the exact conv output: device {rel(zd, z64):.2e}   cpu-fp32 {rel(z32, z64):.2e}")
# ConvTranspose
for (n, ci, co, h, w) in ((1, 512, 256, 24, 36), (1, 256, 128, 48, 72)):
    x = F.silu(torch.randn(n, ci, h, w, generator=g))
    wt = torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    y64 = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2)
    y32 = F.conv_transpose2d(x, wt, b, stride=2)
    yd = ops.convT2x2_fwd(x.cuda(), wt.cuda(), b.cuda())
    print(f"convT {ci}->{co} {h}x{w}: device {rel(yd, y64):.2e}   cpu-fp32 {rel(y32, y64):.2e}")
# weight gradient
for (n, ci, co, h, w) in ((6, 128, 128, 96, 144), (6, 64, 64, 192, 288), (2, 512, 256, 48, 72)):
    x = F.silu(torch.randn(n, ci, h, w, generator=g))
    dy = torch.randn(n, co, h, w, generator=g)
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wt, padding=1) * dy.double()).sum().backward()
    w32 = torch.zeros(co, ci, 3, 3, requires_grad=True)
    (F.conv2d(x, w32, padding=1) * dy).sum().backward()
    print(f"wgrad N{n} {ci}->{co} {h}x{w}: cpu-fp32 {rel(w32.grad, wt.grad):.2e}")
    for name, base in (("fp32", 0), ("bf16x6", ops.SPLIT_BASE), ("fp16x3", ops.H3_BASE)):
        for u in (2, 8):
            gbuf = torch.zeros(co, 9, ci, device="cuda")
            ops.wgrad3x3(x.cuda(), dy.cuda(), gbuf, config=base + (u << 8))
            dw = ops.wgrad3x3_unpack(gbuf)
            print(f"     {name:7s} rounds {u}: {rel(dw, wt.grad):.2e}")
