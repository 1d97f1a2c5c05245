"""Where does the EAGER side-stream schedule (weight gradients on child streams beside two micro-batches: four streams)
differ from the serial one?  One eager step each on identical parameters and inputs, REPS times; compared: the loss, the
forward's saved tensors (bit for bit), and every gradient tensor (relative to the step's gradient scale).

    python tools/overlap_race_probe.py [--reps 5] [--out gpurun_out/overlap_race_probe.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd import engine  # noqa: E402
from climate_amd.config import synthetic_config  # noqa: E402
from climate_amd.model import get_model  # noqa: E402
from climate_amd.trainer import HotPathTrainer  # noqa: E402


def one_step(overlap, x, y, cfg, micro):
    engine.OVERLAP_WGRAD = overlap
    torch.manual_seed(cfg.seed)
    m = get_model(cfg).cuda()
    tr = HotPathTrainer(m, lr=0.0, use_graph=False, distributed=False, micro_batches=micro)
    tr.keep_saved = True
    tr._fwd_bwd(x, y)
    torch.cuda.synchronize()
    sv = tr.saved if isinstance(tr.saved, (list, tuple)) else [tr.saved]
    fwd = {}
    for h, s in enumerate(sv):
        for i, c in enumerate(s.enc):
            fwd[f"half{h}.enc{i + 1}.out"] = c.out.clone()
            fwd[f"half{h}.enc{i + 1}.fmap"] = c.fmap.clone()
        for name, (c, _x) in zip(("up3", "up2", "up1"), s.ups):
            fwd[f"half{h}.{name}.out"] = c.out.clone()
        fwd[f"half{h}.lstm.bott"] = s.lstm.bott.clone()
    grads = {k: v.clone() for k, v in m._views(tr.grad).items()}
    return tr.loss.item(), fwd, grads


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--micro", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "overlap_race_probe.txt"))
    args = ap.parse_args()
    cfg = synthetic_config(base_channels=32, seq_len=6)
    gen = torch.Generator("cpu").manual_seed(7)
    x = torch.randn(32, 6, 5, 48, 72, generator=gen).cuda()
    y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        def say(*a):
            line = " ".join(str(v) for v in a)
            print(line)
            f.write(line + "\n")
            f.flush()
        one_step(False, x, y, cfg, args.micro)                      # autotune everything first
        l0, f0, g0 = one_step(False, x, y, cfg, args.micro)
        big = max(v.norm().item() for v in g0.values())
        for what in ("serial", "overlap"):
            for rep in range(args.reps):
                l1, f1, g1 = one_step(what == "overlap", x, y, cfg, args.micro)
                bad_f = [k for k in f0 if not torch.equal(f0[k], f1[k])]
                errs = sorted(((g1[k] - g0[k]).norm().item() / max(g0[k].norm().item(), 1e-3 * big), k) for k in g0)[::-1]
                worst = ", ".join(f"{k} {e:.1e}" for e, k in errs[:4])
                say(f"{what:8s} rep {rep}: loss diff {abs(l1 - l0) / abs(l0):.1e}; forward tensors differing: {len(bad_f)} "
                    f"{bad_f[:3]}; worst gradients: {worst}")
