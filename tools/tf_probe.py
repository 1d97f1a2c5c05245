"""Diagnostic: cnn_transformer trainer-path gradients / Adam moments vs the fp64 oracle, step by step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F

import oracle
from climate_amd.cnn_transformer import CNNTransformer
from climate_amd.trainer import HotPathTrainer

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()

torch.manual_seed(3)
m = CNNTransformer(5, 2, 256, 2, 8, 256, dropout=0.0)
P = {k: v.detach().clone() for k, v in m.state_dict().items()}
gen = torch.Generator("cpu").manual_seed(4)
x = torch.randn(3, 5, 48, 72, generator=gen); y = torch.randn(3, 2, 48, 72, generator=gen)
pf = {k: v.double().clone().requires_grad_() for k, v in P.items()}
opt = torch.optim.Adam(list(pf.values()), lr=5e-4)
for graph in (False, True):
    pf = {k: v.double().clone().requires_grad_() for k, v in P.items()}
    opt = torch.optim.Adam(list(pf.values()), lr=5e-4)
    m2 = CNNTransformer(5, 2, 256, 2, 8, 256, dropout=0.0); m2.load_state_dict(P)
    tr = HotPathTrainer(m2.cuda(), lr=5e-4, use_graph=graph, distributed=False)
    names = [n for n, _ in m2.named_parameters()]
    for step in range(3):
        opt.zero_grad()
        l = F.mse_loss(oracle.cnn_transformer_forward(pf, x.double(), 8), y.double()); l.backward()
        # device gradient at the ORACLE's current parameters would need a reload; instead report drift
        opt.step()
        # oracle gradient at the DEVICE's current parameters
        pd = {k: v.detach().cpu().double().clone().requires_grad_() for k, v in m2.state_dict().items()}
        F.mse_loss(oracle.cnn_transformer_forward(pd, x.double(), 8), y.double()).backward()
        lh = tr.step(x.cuda(), y.cuda()).item()
        gdev = tr.grad.clone()
        off = 0
        grow = []
        for k, prm in m2.named_parameters():
            n = prm.numel()
            grow.append((rel(gdev[off:off + n].view(prm.shape), pd[k].grad), k)); off += n
        grow.sort(reverse=True)
        oo = sorted(((rel(pd[k].grad, pf[k].grad), k) for k in pd), reverse=True)
        print("   oracle grad at device params vs at oracle params:", " | ".join(f"{a:.1e} {b}" for a, b in oo[:4]))
        print("   grad vs oracle at device params:", " | ".join(f"{a:.1e} {b}" for a, b in grow[:8]))
        st = tr.optimizer_state_dict()["state"]
        sd = m2.state_dict()
        rows = []
        dd = sorted(((((sd[k].cpu().double() - pf[k].detach()).abs().max()).item(), int(((sd[k].cpu().double() - pf[k].detach()).abs() > 1e-5).sum()), k) for k in names), reverse=True)
        print("   max |p_dev - p_oracle| (count > 1e-5):", " | ".join(f"{a:.1e} ({c}) {b}" for a, c, b in dd[:6]))
        for i, k in enumerate(names):
            rows.append((rel(st[i]["exp_avg"], opt.state[pf[k]]["exp_avg"]), rel(sd[k], pf[k]), k))
        rows.sort(reverse=True)
        print(f"graph={graph} step {step} loss {lh:.7f} vs {l.item():.7f}")
        for r in rows[:3]:
            print(f"   m rel {r[0]:.2e}  p rel {r[1]:.2e}  {r[2]}")
        rows.sort(key=lambda r: -r[1])
        for r in rows[:6]:
            print(f"   P rel {r[1]:.2e}  m rel {r[0]:.2e}  {r[2]}")
