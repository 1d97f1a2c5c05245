"""Debug aid (r01): eager overlap produced non-finite gradients.  Records every op output of the second backward
WITHOUT synchronising, then reports the first non-finite one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd import ops
from climate_amd.config import synthetic_config
from climate_amd.model import get_model
from climate_amd.trainer import HotPathTrainer
cfg = synthetic_config(base_channels=32, seq_len=6)
gen = torch.Generator("cpu").manual_seed(7)
x = torch.randn(32, 6, 5, 48, 72, generator=gen).cuda()
y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()
torch.manual_seed(cfg.seed)
m = get_model(cfg).cuda()
tr = HotPathTrainer(m, lr=1e-3, use_graph=False)
log = []
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        log.append((name, [t for t in outs if torch.is_tensor(t)], [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]))
        return out
    setattr(ops, name, w)
for n_ in ("gates_bwd", "gn_silu_bwd_gated", "gn_silu_bwd", "conv3x3", "maxpool2_bwd", "convT2x2_bwd", "head_mse_bwd",
           "se_spatial_gate_fwd", "gn_silu_fwd", "time_mean", "convT2x2_fwd"):
    wrap(n_)
for it in range(2):
    log.clear()
    tr._fwd_bwd(x, y)
    torch.cuda.synchronize()
    g = m._views(tr.grad)
    bad = [k for k, v in g.items() if not torch.isfinite(v).all()]
    print("iter", it, "loss", tr.loss.item(), "non-finite grads:", len(bad))
    for i, (name, outs, shp) in enumerate(log):
        nf = [j for j, t in enumerate(outs) if not torch.isfinite(t).all()]
        if name == "gates_bwd" and (outs[1] == 0).any():
            print(f"   op #{i} gates_bwd: cnt == 0 at {(outs[1] == 0).sum().item()} pixels (tie count lost: the tie test "
                  f"saw different a2 / s / max values than the forward stored)")
        if nf:
            print(f"   first non-finite output: op #{i} {name} outputs {nf} arg shapes {shp}; previous ops:",
                  [l[0] for l in log[max(0, i - 4):i]])
            break
