"""What does a cross-stream dependency edge cost inside a replayed hipGraph on this runtime?  Three graphs of the same 64 small
launches (cm_zero of 1 MB): all on one stream; ping-pong between two streams (every launch waits for the previous one on the
other stream: 64 cross-stream edges, no concurrency possible); and a fork / join pair around every second launch (the
fine-grained side-stream pattern of the weight-gradient overlap).  Reported: microseconds per replay and per edge.

    python tools/graph_edge_cost.py [--out gpurun_out/graph_edge_cost.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd._lib import check, lib  # noqa: E402

N = 64


def launch(buf):
    check(lib.cm_zero(buf.data_ptr(), buf.numel() * 4, torch.cuda.current_stream().cuda_stream), "zero")


def timed(g, replays=50):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / replays


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "graph_edge_cost.txt"))
    args = ap.parse_args()
    bufs = [torch.empty(1 << 18, device="cuda") for _ in range(2)]
    side = torch.cuda.Stream()
    launch(bufs[0])
    torch.cuda.synchronize()

    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        for i in range(N):
            launch(bufs[0])

    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        main = torch.cuda.current_stream()
        for i in range(N // 2):
            launch(bufs[0])
            side.wait_stream(main)
            with torch.cuda.stream(side):
                launch(bufs[0])
            main.wait_stream(side)

    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        main = torch.cuda.current_stream()
        for i in range(N // 2):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                launch(bufs[1])              # independent work on the side stream ...
            launch(bufs[0])                  # ... beside this launch
        main.wait_stream(side)

    t1, t2, t3 = timed(g1), timed(g2), timed(g3)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for line in (
            f"{N} launches on one stream:                                   {t1:8.1f} us per replay ({t1 / N:.2f} us per launch)",
            f"{N} launches ping-pong between two streams ({N} cross edges):   {t2:8.1f} us per replay "
            f"(+{(t2 - t1) / N:.2f} us per cross-stream edge)",
            f"{N // 2} launches + {N // 2} forked side launches ({N // 2} fork edges, 1 join): {t3:8.1f} us per replay "
            f"(+{(t3 - t1 / 2) / (N // 2):.2f} us per fork over {N // 2} serial launches)",
        ):
            print(line)
            f.write(line + "\n")
