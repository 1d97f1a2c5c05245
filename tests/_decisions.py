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
    u = ctx.activation2() * ctx.s[:, :, None, None]
    return u == ctx.fmap[:, 1:2]


def hip_decisions(sv, delta=1e-5):
    """sv: engine.Saved of one forward -- or the list of the micro-batches' Saved objects (contiguous parts of the batch,
    in order: HotPathTrainer.saved with micro_batches=2).  Returns oracle.Decisions keyed like the oracle's sites."""
    if isinstance(sv, (list, tuple)):
        parts = [hip_decisions(q, delta) for q in sv]
        dec = oracle.Decisions(delta=delta)
        for k in parts[0].amax:
            dec.amax[k] = torch.cat([q.amax[k] for q in parts], 0)
        for k in parts[0].pool:
            dec.pool[k] = torch.cat([q.pool[k] for q in parts], 0)
        return dec
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


def transformer_relu_decisions(sv, delta=1e-5):
    """cnn_transformer: the device path's ReLU on/off decisions (post-ReLU activations it saved, > 0) in the oracle's
    layouts.  sv: climate_amd.cnn_transformer._Saved of a dropout-free forward."""
    dec = oracle.Decisions(delta=delta)
    B, _cin, H, W = sv.shape
    e2, e = sv.y1.shape[1], sv.t0.shape[1]
    dec.relu["encoder.0"] = (sv.y1 > 0).view(B, H // 2, W // 2, e2).permute(0, 3, 1, 2).cpu()     # token-major -> NCHW
    dec.relu["encoder.2"] = (sv.t0 > 0).view(B, H // 4, W // 4, e).permute(0, 3, 1, 2).cpu()
    for i, layer in enumerate(sv.layers):
        h1 = layer[7]
        dec.relu[("mlp", i)] = (h1 > 0).view(B, -1, h1.shape[1]).cpu()
    dec.relu["decoder.0"] = (sv.dec1 > 0).cpu()
    dec.relu["decoder.2"] = (sv.dec2 > 0).cpu()
    return dec
