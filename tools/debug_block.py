#!/usr/bin/env python3
"""Op-by-op comparison of one ConvBlock (forward + backward) against torch fp64 at an arbitrary shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from climate_amd import ops

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()

n, ci, co, h, w = map(int, sys.argv[1:6])
g = torch.Generator("cpu").manual_seed(5)
x = torch.randn(n, ci, h, w, generator=g)
dout = torch.randn(n, co, h, w, generator=g) * 1e-5
W1 = torch.randn(co, ci, 3, 3, generator=g) * (9 * ci) ** -0.5
W2 = torch.randn(co, co, 3, 3, generator=g) * (9 * co) ** -0.5
g1 = 1 + 0.1 * torch.randn(co, generator=g); b1 = 0.1 * torch.randn(co, generator=g)
g2 = 1 + 0.1 * torch.randn(co, generator=g); b2 = 0.1 * torch.randn(co, generator=g)
S1 = torch.randn(co // 8, co, 1, 1, generator=g) * 0.3; S2 = torch.randn(co, co // 8, 1, 1, generator=g) * 0.3
W7 = torch.randn(1, 2, 7, 7, generator=g) * 0.1
P = dict(W1=W1, W2=W2, g1=g1, b1=b1, g2=g2, b2=b2, S1=S1, S2=S2, W7=W7)
R = {k: v.double().clone().requires_grad_() for k, v in P.items()}
xd = x.double().requires_grad_()
y1 = F.conv2d(xd, R["W1"], padding=1); a1 = F.silu(F.group_norm(y1, 8, R["g1"], R["b1"], 1e-5))
y2 = F.conv2d(a1, R["W2"], padding=1); a2 = F.silu(F.group_norm(y2, 8, R["g2"], R["b2"], 1e-5))
pooled = a2.mean((2, 3), keepdim=True); z = F.conv2d(pooled, R["S1"]); s = torch.sigmoid(F.conv2d(F.relu(z), R["S2"]))
U = a2 * s; fmap = torch.cat([U.mean(1, keepdim=True), U.amax(1, keepdim=True)], 1)
gate = torch.sigmoid(F.conv2d(fmap, R["W7"], padding=3)); out = U * gate
for t in (y1, a1, y2, a2, pooled, s, U, fmap, gate): t.retain_grad()
out.backward(dout.double())
D = lambda t: t.cuda().contiguous()
Y1 = ops.conv3x3(D(x), ops.pack_conv3x3(D(W1)), co)
A1, st1, _ = ops.gn_silu_fwd(Y1, D(g1), D(b1))
Y2 = ops.conv3x3(A1, ops.pack_conv3x3(D(W2)), co)
A2, st2, PO = ops.gn_silu_fwd(Y2, D(g2), D(b2), want_pooled=True)
Z, S = ops.se_excite_fwd(PO, D(S1), D(S2))
OUT, FM, GT = ops.spatial_gate_fwd(A2, S, D(W7))
print("fwd: y1 %.1e a1 %.1e y2 %.1e a2 %.1e pooled %.1e s %.1e fmap %.1e gate %.1e out %.1e" % (
    rel(Y1, y1), rel(A1, a1), rel(Y2, y2), rel(A2, a2), rel(PO, pooled.flatten(1)), rel(S, s.flatten(1)), rel(FM, fmap),
    rel(GT, gate.squeeze(1)), rel(OUT, out)))
G = {k: torch.zeros_like(D(v)) for k, v in P.items()}
dmap, (umax, cnt), dpool = ops.gates_bwd(D(dout), A2, S, Z, PO, GT, FM, D(S1), D(S2), D(W7), G["S1"], G["S2"], G["W7"])
print("bwd maps: dmap %.1e dpool %.1e  dS1 %.1e dS2 %.1e dW7 %.1e" % (rel(dmap, fmap.grad), rel(dpool, pooled.grad.flatten(1)),
      rel(G["S1"], R["S1"].grad), rel(G["S2"], R["S2"].grad), rel(G["W7"], R["W7"].grad)))
DY2 = ops.gn_silu_bwd_gated(Y2, D(g2), D(b2), st2, A2, D(dout), GT, dmap, umax, cnt, S, dpool, G["g2"], G["b2"])
print("gn2 bwd: dy2 %.1e dg2 %.1e db2 %.1e" % (rel(DY2, y2.grad), rel(G["g2"], R["g2"].grad), rel(G["b2"], R["b2"].grad)))
gw = torch.zeros(co, 9, co, device="cuda"); ops.wgrad3x3(A1, DY2, gw)
DA1 = ops.conv3x3(DY2, ops.pack_conv3x3(D(W2), dgrad=True), co)
print("conv2 bwd: dW2 %.1e da1 %.1e" % (rel(ops.wgrad3x3_unpack(gw), R["W2"].grad), rel(DA1, a1.grad)))
DY1 = ops.gn_silu_bwd(Y1, D(g1), D(b1), st1, DA1, G["g1"], G["b1"])
print("gn1 bwd: dy1 %.1e dg1 %.1e db1 %.1e" % (rel(DY1, y1.grad), rel(G["g1"], R["g1"].grad), rel(G["b1"], R["b1"].grad)))
gw = torch.zeros(co, 9, ci, device="cuda"); ops.wgrad3x3(D(x), DY1, gw)
print("conv1 bwd: dW1 %.1e" % rel(ops.wgrad3x3_unpack(gw), R["W1"].grad))

# ---- deeper: ds = dL/ds per (n,c), and the pieces se_bwd_reduce consumes
from climate_amd._lib import lib, check
st = torch.cuda.current_stream().cuda_stream
hw = h * w
dgpre = torch.empty(n, h, w, device="cuda"); cnt2 = torch.empty(n, h, w, device="cuda"); um2 = torch.empty(n, h, w, device="cuda")
check(lib.cm_gate_bwd_reduce(D(dout).data_ptr(), A2.data_ptr(), S.data_ptr(), GT.data_ptr(), dgpre.data_ptr(), cnt2.data_ptr(), um2.data_ptr(), n, co, hw, st))
dmap2 = torch.empty(n, 2, h, w, device="cuda"); dw7 = torch.zeros(98, device="cuda")
check(lib.cm_conv7_bwd(dgpre.data_ptr(), FM.data_ptr(), D(W7).data_ptr(), dmap2.data_ptr(), dw7.data_ptr(), torch.empty(int(lib.cm_conv7_bwd_scratch_elems(n, h)), device='cuda').data_ptr(), n, h, w, st))
ds = torch.empty(n, co, device="cuda")
check(lib.cm_se_bwd_reduce(D(dout).data_ptr(), A2.data_ptr(), S.data_ptr(), GT.data_ptr(), dmap2.data_ptr(), um2.data_ptr(), cnt2.data_ptr(), ds.data_ptr(), n, co, hw, None, 0, None, st))
print("ds %.2e  cnt max %d  dgpre %.1e" % (rel(ds, s.grad.flatten(1)), int(cnt2.max().item()), rel(dgpre, (fmap.grad * 0 + 0).sum(1) if False else dgpre)))
dsr = s.grad.flatten(1)
print("per (n,c) rel:", [f"{((ds[i//co, i%co].item()-dsr[i//co,i%co].item())/abs(dsr[i//co,i%co].item())):.1e}" for i in range(min(n*co, 32))])
# same reduction on the host in fp64 from the HIP inputs
Ug = (A2.double().cpu() * S.double().cpu()[:, :, None, None])
dU = D(dout).double().cpu() * GT.double().cpu()[:, None] + dmap2.double().cpu()[:, 0:1] / co + dmap2.double().cpu()[:, 1:2] * (Ug == FM.double().cpu()[:, 1:2]).double() / cnt2.double().cpu()[:, None]
ds_host = (dU * A2.double().cpu()).sum((2, 3))
print("ds(HIP kernel) vs fp64 host sum of HIP inputs: %.2e ; host-sum vs torch ref: %.2e" % (rel(ds, ds_host), rel(ds_host, dsr)))
z64 = z.flatten(1); print("SE hidden z (fp64):", z64.detach().numpy().round(6).tolist())
Ut = A2 * S[:, :, None, None]
eq = (Ut == FM[:, 1:2])
print("pixels with no channel equal to stored max:", int((eq.sum(1) == 0).sum().item()), "of", n * hw,
      "; cnt kernel==torch:", bool((cnt2 == eq.sum(1).float()).all().item()), "; cnt min", cnt2.min().item())
mxt = Ut.amax(1)
print("stored max == torch amax of product everywhere:", bool((mxt == FM[:, 1]).all().item()),
      " max abs diff", (mxt - FM[:, 1]).abs().max().item())
dUt = D(dout).double() * GT.double()[:, None] + dmap2.double()[:, 0:1] / co + dmap2.double()[:, 1:2] * eq.double() / cnt2.double()[:, None]
ds_t = (dUt * A2.double()).sum((2, 3))
print("kernel ds vs fp64-sum of HIP inputs: %.2e ; that fp64-sum vs torch ref: %.2e" % (rel(ds, ds_t), rel(ds_t, dsr)))
print("dmap ch0 %.2e ch1 %.2e ; fmap.grad norms %.3e %.3e" % (rel(dmap2[:, 0], fmap.grad[:, 0]), rel(dmap2[:, 1], fmap.grad[:, 1]), fmap.grad[:, 0].norm().item(), fmap.grad[:, 1].norm().item()))
i = 21; nn_, cc_ = i // co, i % co
t1 = (D(dout).double()[nn_, cc_] * GT.double()[nn_] * A2.double()[nn_, cc_]).sum().item()
t2 = (dmap2.double()[nn_, 0] / co * A2.double()[nn_, cc_]).sum().item()
t3 = (dmap2.double()[nn_, 1] * eq.double()[nn_, cc_] / cnt2.double()[nn_] * A2.double()[nn_, cc_]).sum().item()
print("entry (n=%d,c=%d): terms dout*gate %.4e, da/C %.4e, dm*tie %.4e ; kernel %.6e ref %.6e ; #max pixels %d" % (nn_, cc_, t1, t2, t3, ds[nn_, cc_].item(), dsr[nn_, cc_].item(), int(eq[nn_, cc_].sum().item())))
eq_ref = (U.detach() == fmap.detach()[:, 1:2]).cuda()
mism = (eq_ref != eq)
print("argmax pattern mismatches HIP(fp32) vs fp64 ref: %d pixels-channel entries; per channel:" % int(mism.sum().item()), mism.sum((0, 2, 3)).tolist())
# gap between top-2 at mismatching pixels
top2 = Ut.topk(2, dim=1).values
gap = ((top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30))
mp = mism.any(1)
print("relative top-2 gap at mismatching pixels:", gap[mp][:10].tolist())
# is U structurally tied? distribution of gap
print("fraction of pixels with rel gap < 1e-6: %.2e ; < 1e-4: %.2e" % ((gap.abs() < 1e-6).float().mean().item(), (gap.abs() < 1e-4).float().mean().item()))
