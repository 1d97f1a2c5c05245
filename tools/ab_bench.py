"""Same-box A/B of schedule switches at BASELINE config 2 through the graphed trainer: every variant gets its own trainer
(same initial parameters), the variants are timed round-robin (ABCABC...) so that box-to-box and drift effects cancel.

    python tools/ab_bench.py [--rounds 3] [--steps 30] [--out gpurun_out/ab.txt] [variant ...]

Variants: base | tail_off | tail_min<N> | lstm_step_off | overlap_wgrad | micro1 | gn_epi_off
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd import engine, ops  # noqa: E402
from climate_amd.config import synthetic_config  # noqa: E402
from climate_amd.model import get_model  # noqa: E402
from climate_amd.trainer import HotPathTrainer  # noqa: E402

DEFAULTS = dict(tail=ops.BLOCK_TAIL, tail_min=ops.BLOCK_TAIL_MIN_N, lstm=ops.LSTM_STEP, ow=engine.OVERLAP_WGRAD,
                gn=ops.GN_EPILOGUE, lstmb=ops.LSTM_STEP_BWD)


def apply(v):
    ops.BLOCK_TAIL, ops.BLOCK_TAIL_MIN_N, ops.LSTM_STEP, engine.OVERLAP_WGRAD = (DEFAULTS["tail"], DEFAULTS["tail_min"],
                                                                                 DEFAULTS["lstm"], DEFAULTS["ow"])
    ops.GN_EPILOGUE = DEFAULTS["gn"]
    ops.LSTM_STEP_BWD = DEFAULTS["lstmb"]
    micro = None
    if v == "tail_off":
        ops.BLOCK_TAIL = False
    elif v.startswith("tail_min"):
        ops.BLOCK_TAIL_MIN_N = int(v[len("tail_min"):])
    elif v == "lstm_step_off":
        ops.LSTM_STEP = False
    elif v == "overlap_wgrad":
        engine.OVERLAP_WGRAD = True
    elif v == "micro1":
        micro = 1
    elif v == "lstm_bwd_off":
        ops.LSTM_STEP_BWD = False
    elif v == "gn_epi_off":
        ops.GN_EPILOGUE = False
    elif v != "base":
        raise SystemExit(f"unknown variant {v}")
    return micro


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", default=["base", "tail_off", "tail_min32", "lstm_step_off", "overlap_wgrad"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ab.txt"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synthetic_config(base_channels=32, seq_len=6)
    gen = torch.Generator("cpu").manual_seed(1234)
    x = torch.randn(32, 6, 5, 48, 72, generator=gen).to(dev)
    y = torch.randn(32, 2, 48, 72, generator=gen).to(dev)
    trainers = {}
    for v in args.variants:
        micro = apply(v)
        torch.manual_seed(cfg.seed)
        m = get_model(cfg).to(dev)
        tr = HotPathTrainer(m, use_graph=True, distributed=False, micro_batches=micro)
        for _ in range(5):
            tr.step(x, y)                      # capture under this variant's switches + warm-up
        torch.cuda.synchronize()
        trainers[v] = tr
    res = {v: [] for v in args.variants}
    for _ in range(args.rounds):
        for v in args.variants:
            tr = trainers[v]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tr.step(x, y)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / args.steps * 1e3)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for v in args.variants:
            ms = sorted(res[v])[len(res[v]) // 2]
            line = f"{v:16s} median {ms:.3f} ms/step = {32 / ms * 1e3:7.1f} samples/s   rounds: " + " ".join(f"{t:.3f}" for t in res[v])
            print(line)
            f.write(line + "\n")
