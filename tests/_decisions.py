"""Test aid: read the DEVICE path's discrete choices (amax tie masks, MaxPool argmax) out of the engine's saved
activations so that the fp64 oracle can impose -- and validate -- them (oracle.Decisions)."""
import torch

import oracle


def _pool_idx(x):
    """first maximal element (row-major) of every 2x2 window of x [N,C,H,W]: what cm_maxpool2_bwd routes to."""
    n, c, h, w = x.shape
    win = x.view(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    return win.argmax(-1)          # torch.argmax returns the FIRST maximal index


def _mask(ctx):
    """channels attaining the stored channel maximum of U = a2 * s (fp32 product, same operands as the kernels)."""
    u = ctx.a2 * ctx.s[:, :, None, None]
    return u == ctx.fmap[:, 1:2]


def hip_decisions(sv, delta=1e-5):
    """sv: engine.Saved of one forward.  Returns oracle.Decisions keyed like the oracle's sites."""
    dec = oracle.Decisions(delta=delta)
    T = sv.T
    enc_prefix = ("enc1.", "enc2.conv.", "enc3.conv.", "enc4.conv.")
    pool_prefix = (None, "enc2.", "enc3.", "enc4.")
    for i, ctx in enumerate(sv.enc):
        m = _mask(ctx).cpu()
        for t in range(T):
            dec.amax[(enc_prefix[i], t)] = m[t::T].contiguous()        # folded batch: sample n = b*T + t
        if i < 3:
            idx = _pool_idx(ctx.out).cpu()
            for t in range(T):
                dec.pool[(pool_prefix[i + 1], t)] = idx[t::T].contiguous()
    for name, (ctx, _x) in zip(("up3.conv.", "up2.conv.", "up1.conv."), sv.ups):
        dec.amax[(name, None)] = _mask(ctx).cpu()
    return dec
