"""Localise the device-vs-oracle gradient discrepancy at a saved cnn_transformer parameter state (tools/repro/tf_bad_state.pt):
walk the decoder tail of the backward -- gradient wrt the head input, ReLU masks, transposed-conv data gradients."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from climate_amd import ops  # noqa: E402
from climate_amd import cnn_transformer as ct  # noqa: E402


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


P = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "repro", "tf_bad_state.pt"), weights_only=True)
gen = torch.Generator("cpu").manual_seed(4)
x = torch.randn(3, 5, 48, 72, generator=gen); y = torch.randn(3, 2, 48, 72, generator=gen)
pdev = {k: v.cuda() for k, v in P.items()}
pred, sv = ct.forward(pdev, x.cuda(), 8, save=True, head=True)
# float64 reference of the tail from the DEVICE's z (transformer output), so only the tail is compared
z = sv.z.detach().cpu().double().requires_grad_()
pd = {k: v.double() for k, v in P.items()}
d1 = F.relu(F.conv_transpose2d(z, pd["decoder.0.weight"], pd["decoder.0.bias"], stride=2)); d1.retain_grad()
a2 = F.conv_transpose2d(d1, pd["decoder.2.weight"], pd["decoder.2.bias"], stride=2); a2.retain_grad()
d2 = F.relu(a2); d2.retain_grad()
pr = F.conv2d(d2, pd["decoder.4.weight"], pd["decoder.4.bias"])
loss = F.mse_loss(pr, y.double()); loss.backward()
print("forward: d1", rel(sv.dec1, d1), " d2", rel(sv.dec2, d2), " pred", rel(pred, pr))
dpred = (2.0 / pred.numel()) * (pred - y.cuda())
gW = torch.zeros_like(pdev["decoder.4.weight"]); gb = torch.zeros_like(pdev["decoder.4.bias"])
dd_head = ops.head_bwd(dpred, sv.dec2, pdev["decoder.4.weight"], gW, gb)
print("d(d2) from head_bwd          ", rel(dd_head, d2.grad))
lossbuf = torch.zeros(1, device="cuda"); gW2 = torch.zeros_like(gW); gb2 = torch.zeros_like(gb)
dd_f = ops.head_mse_bwd(sv.dec2, pdev["decoder.4.weight"], pdev["decoder.4.bias"], y.cuda(), lossbuf, gW2, gb2)
print("d(d2) from fused head_mse_bwd", rel(dd_f, d2.grad), " dW", rel(gW2, gW), " loss", lossbuf.item(), loss.item())
dd2 = ops.relu_mask_(dd_f.clone(), sv.dec2)
print("after ReLU mask (d a2)       ", rel(dd2, a2.grad))
m_dev = (sv.dec2 > 0).cpu(); m_ref = a2.detach() > 0
print("   mask disagreements:", int((m_dev != m_ref).sum()), "of", m_ref.numel(),
      "; |grad| carried by them:", (d2.grad[m_dev != m_ref]).abs().sum().item() if (m_dev != m_ref).any() else 0.0)
g2w = torch.zeros_like(pdev["decoder.2.weight"]); g2b = torch.zeros_like(pdev["decoder.2.bias"])
dd1 = ops.convT2x2_bwd(sv.dec1, pdev["decoder.2.weight"], dd2, g2w, g2b)
print("decoder.2: d(d1)", rel(dd1, d1.grad), " bias grad", rel(g2b, a2.grad.sum((0, 2, 3))),
      " weight grad", rel(g2w, torch.autograd.grad(F.mse_loss(F.conv2d(F.relu(F.conv_transpose2d(d1.detach(), pd["decoder.2.weight"].requires_grad_(), pd["decoder.2.bias"], stride=2)), pd["decoder.4.weight"], pd["decoder.4.bias"]), y.double()), pd["decoder.2.weight"])[0]))
