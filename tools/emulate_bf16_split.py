#!/usr/bin/env python3
"""CPU emulation of split-bf16 convolutions (bf16x3 / bf16x6 on the bf16 matrix cores) inside the oracle, to measure
the parity margin against the fp64 oracle before writing such kernels.  Forward conv and data-gradient conv use the
split operands; the weight gradient stays exact (fp32 MFMA path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import oracle, oracle.cpu_ref as ref
_conv = F.conv2d   # the real one (F.conv2d itself gets patched below)

def split(t, pieces):
    out, r = [], t.double()
    for _ in range(pieces):
        h = r.float().bfloat16().double()
        out.append(h); r = r - h
    return out

def conv_split(x, w, pieces, terms, **kw):
    xs, ws = split(x, pieces), split(w, pieces)
    acc = 0
    for (i, j) in terms:
        acc = acc + _conv(xs[i], ws[j], **kw)
    return acc

TERMS = {"x3": (2, [(0, 0), (0, 1), (1, 0)]), "x4": (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
         "x6": (3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)])}

class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, mode):
        ctx.save_for_backward(x, w); ctx.mode = mode; ctx.hasb = b is not None
        pieces, terms = TERMS[mode]
        y = conv_split(x, w, pieces, terms, padding=1).to(x.dtype)
        return y + b.view(1, -1, 1, 1) if b is not None else y
    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        pieces, terms = TERMS[ctx.mode]
        wt = w.flip(2, 3).transpose(0, 1)
        dx = conv_split(dy, wt, pieces, terms, padding=1).to(x.dtype)
        dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1).to(w.dtype)
        db = dy.sum((0, 2, 3)) if ctx.hasb else None
        return dx, dw, db, None

def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()

def run(mode, base, T, B, H, W, salt=0):
    P = oracle.closed_form_params(5, 2, base, salt=salt)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, T, 5, H, W, generator=g); y = torch.randn(B, 2, H, W, generator=g)
    p64 = {k: v.double().clone().requires_grad_() for k, v in P.items()}
    l64 = oracle.training_loss(p64, x.double(), y.double()); l64.backward()
    orig = F.conv2d
    def patched(inp, wt, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if wt.shape[-1] == 3 and padding == 1 and mode != "fp32":
            return SplitConv.apply(inp, wt, bias, mode)
        return orig(inp, wt, bias, stride, padding, dilation, groups)
    ref.F.conv2d = patched
    try:
        p32 = {k: v.clone().requires_grad_() for k, v in P.items()}
        l32 = oracle.training_loss(p32, x, y); l32.backward()
    finally:
        ref.F.conv2d = orig
    errs = sorted(((rel(p32[k].grad, p64[k].grad), k) for k in p64 if p64[k].grad is not None), reverse=True)
    print(f"{mode:5s} base={base} T={T} B={B} {H}x{W}: loss rel {abs(l32.item()-l64.item())/abs(l64.item()):.1e}  worst grads:",
          ", ".join(f"{e:.1e} {k}" for e, k in errs[:3]), f" median {errs[len(errs)//2][0]:.1e}")

if __name__ == "__main__":
    for shape in ((8, 3, 2, 16, 24), (16, 4, 2, 24, 40), (32, 3, 2, 48, 72)):
        for mode in ("fp32", "x3", "x4", "x6"):
            run(mode, *shape)
