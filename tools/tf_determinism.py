"""Run-to-run determinism of the cnn_transformer fused step (config-4 widths): N fresh (model, trainer) pairs from the
same seed, three steps each; gradients of every step compared with the first repeat, tensor by tensor."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climate_amd.cnn_transformer import CNNTransformer  # noqa: E402
from climate_amd.trainer import HotPathTrainer  # noqa: E402

graph = os.environ.get("TF_GRAPH", "1") != "0"
gen = torch.Generator("cpu").manual_seed(4)
x = torch.randn(3, 5, 48, 72, generator=gen); y = torch.randn(3, 2, 48, 72, generator=gen)
ref = None
for rep in range(int(os.environ.get("TF_REPS", "8"))):
    torch.manual_seed(3)
    m = CNNTransformer(5, 2, 256, 2, 8, 256, dropout=0.0).cuda()
    tr = HotPathTrainer(m, lr=5e-4, use_graph=graph, distributed=False)
    grads = []
    for step in range(3):
        tr.step(x.cuda(), y.cuda())
        torch.cuda.synchronize()
        grads.append({k: v.detach().clone() for k, v in m._views(tr.grad).items()})
    if ref is None:
        ref = grads
        continue
    for step in range(3):
        bad = []
        for k in ref[step]:
            a, b = grads[step][k].double(), ref[step][k].double()
            e = ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
            if e > 1e-5:
                bad.append((e, k))
        if bad:
            bad.sort(reverse=True)
            print(f"rep {rep} step {step}: {len(bad)} tensors differ from rep 0; worst: " +
                  ", ".join(f"{k} {e:.1e}" for e, k in bad[:6]))
print("done")
