"""Reference checksums at the per-rank sizes BASELINE.json configs 3 and 5 define (run once in the authoring container):

    python tests/golden/gen_golden_big.py

  cfg3_b32_checksums.npz   base 64, seq_len 12, 48x72, per-rank batch 32   (configs[2]: global 256 over 8 ranks)
  cfg5_b16_checksums.npz   base 64, seq_len 6, 192x288, per-rank batch 16  (configs[4]: global 128 over 8 ranks)

The reference module (imported from /root/reference as in gen_golden.py) is run ONE SAMPLE AT A TIME and the gradients of
the per-sample losses are averaged: the model has no cross-sample term (GroupNorm, SE, gates and the ConvLSTM state are
per-sample, MSELoss is a mean), so this IS the batch step, and it keeps the autograd graph of the 192x288 case (4.3 GB
per sample) inside this container's memory.  Stored: loss, prediction norm + 64 samples, per-tensor gradient norms.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import load_reference, npz  # noqa: E402
import oracle  # noqa: E402


def se_margin(P, x):
    """smallest |z| / max|z| over every SE hidden pre-activation z = W1 mean_hw(x) the forward evaluates"""
    from oracle import cpu_ref as ref
    seen = []
    orig = ref.se_block

    def spy(xx, w1, w2):
        z = F.conv2d(xx.mean(dim=(2, 3), keepdim=True), w1)
        seen.append((z.abs().min() / z.abs().max().clamp_min(1e-30)).item())
        return orig(xx, w1, w2)
    ref.se_block = spy
    try:
        with torch.no_grad():
            for b in range(x.shape[0]):
                oracle.model_forward(P, x[b:b + 1])
    finally:
        ref.se_block = orig
    return min(seen)


def run(name, base, T, B, H, W, seed):
    in_ch, out_ch = 5, 2
    m, _u, _c = load_reference()
    net = m.AttUNetConvLSTM(in_ch=in_ch, out_ch=out_ch, base=base, seq_len=T)
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(B, T, in_ch, H, W, generator=g)
    yt = torch.randn(B, out_ch, H, W, generator=g)
    # closed-form parameter set (oracle.closed_form_params, salt 9, 10, ...) whose SE hidden units all sit clear of their
    # ReLU kink on this input: one unit within rounding of zero flips between two correct fp32 evaluations and moves the
    # SE weight gradient's NORM by 1e-3..1e-2 (tests/test_model_gpu.py::_se_margin does the same selection)
    for salt in range(9, 40):
        params = oracle.closed_form_params(in_ch, out_ch, base, salt=salt)
        if se_margin(params, x) > 2e-3:
            break
    else:
        raise SystemExit("no parameter set with a safe SE ReLU margin")
    print(f"{name}: salt {salt}", flush=True)
    net.load_state_dict(params)
    preds, loss = [], 0.0
    for b in range(B):
        pred = net(x[b:b + 1])
        lb = F.mse_loss(pred, yt[b:b + 1]) / B          # batch MSE = mean of the per-sample MSEs
        lb.backward()                                   # .grad accumulates the batch gradient
        preds.append(pred.detach())
        loss += lb.item()
        print(f"{name}: sample {b + 1}/{B}", flush=True)
    pred = torch.cat(preds, 0)
    idx = torch.linspace(0, pred.numel() - 1, 64).long()
    names, norms = [], []
    for k, p in net.named_parameters():
        if p.grad is not None:
            names.append(k); norms.append(p.grad.double().norm().item())
    npz(name, cfg=np.array([in_ch, out_ch, base, T, B, H, W]), seed=np.array(seed), salt=np.array(salt), loss=np.array(loss),
        pred_l2=pred.double().norm(), pred_samples=pred.flatten()[idx], pred_sample_idx=idx,
        grad_names=np.array(names), grad_l2=np.array(norms),
        how=np.array("one sample at a time through the reference module, gradients of loss_b / B accumulated"))


if __name__ == "__main__":
    torch.set_num_threads(8)
    run("cfg3_b32_checksums.npz", 64, 12, 32, 48, 72, 1234)
    run("cfg5_b16_checksums.npz", 64, 6, 16, 192, 288, 1234)
