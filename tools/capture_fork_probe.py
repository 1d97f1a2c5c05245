"""Probe for the round-2 crash "a fork inside a forked stream under capture segfaults in hipStreamEndCapture".

Each variant runs in a CHILD process (python faulthandler on, so a segfault leaves a traceback instead of taking the
probe down) and the outcome is appended to the report:

  topo-sibling pure torch, three streams, no second-level fork: M forks S and C_M; both join M
  topo-own     pure torch: main stream M forks S (the second micro-batch); M forks its own child C_M, S forks its own
               child C_S; every child joins its parent; S joins M.  (the round-3 topology)
  topo-shared  pure torch: the same, but both M and S fork into and join from ONE child C (what round 2 did: one side
               stream per DEVICE)
  topo-joinorigin  topo-own + the second-level child C_S also joins the origin M directly
  topo-prealloc    topo-own with no allocation on any child stream (outputs pre-allocated)
  topo-prefork     C_S and C_M enter the capture as first-level forks of M at its start; S's work only adds a
                   dependency edge to C_S later; C_S joins M directly (never S)
  topo-preforkjoin topo-prefork + the inner join (S waits for C_S): the topology the trainer records since round 3
  model-own    the trainer at BASELINE config 2, two micro-batches, weight gradients on side streams, graph captured
               and replayed 3 times, losses compared with the one-stream schedule
  model-shared the same with the round-2 stream table (one side stream per device)

    python tools/capture_fork_probe.py [--out gpurun_out/capture_fork_probe.txt]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def topo(how: str):
    """how: own | shared | sibling | joinorigin | prealloc | prefork | preforkjoin (see the module docstring / report)."""
    import torch
    dev = torch.device("cuda", 0)
    a = torch.zeros(1 << 20, device=dev)
    b = torch.zeros(1 << 20, device=dev)
    a2 = torch.zeros(1 << 20, device=dev)
    b2 = torch.zeros(1 << 20, device=dev)
    M = torch.cuda.Stream(dev)
    S = torch.cuda.Stream(dev)
    CM = torch.cuda.Stream(dev)
    CS = CM if how == "shared" else torch.cuda.Stream(dev)
    prealloc = how in ("prealloc", "prefork", "preforkjoin", "sibling")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(M):
        with torch.cuda.graph(g, stream=M):
            S.wait_stream(M)                      # fork the second half
            if how in ("prefork", "preforkjoin"):   # the children enter the capture as FIRST-level forks of the origin ...
                CS.wait_stream(M)
                CM.wait_stream(M)
            with torch.cuda.stream(S):
                b.add_(1.0)
                if how == "sibling":
                    pass                          # (no second-level fork at all: three streams, all forked from M)
                else:
                    CS.wait_stream(S)             # ... and only pick up a dependency edge here (prefork), or fork here
                    with torch.cuda.stream(CS):
                        if prealloc:
                            torch.mul(b, 2.0, out=b2)
                        else:
                            b2 = b * 2.0
                b.add_(1.0)
                if how not in ("sibling", "prefork"):
                    S.wait_stream(CS)             # inner join (preforkjoin: on top of the pre-fork)
            a.add_(1.0)
            CM.wait_stream(M)
            with torch.cuda.stream(CM):
                if prealloc:
                    torch.mul(a, 2.0, out=a2)
                else:
                    a2 = a * 2.0
            a.add_(1.0)
            M.wait_stream(CM)
            M.wait_stream(S)                      # outer join
            if how in ("joinorigin", "prefork", "preforkjoin"):
                M.wait_stream(CS)                 # the second-level child also joins the origin directly
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"topo {how}: captured and replayed; a[0]={a[0].item()} b[0]={b[0].item()} "
          f"a2[0]={a2[0].item()} b2[0]={b2[0].item()}")


def model(shared: bool):
    import torch
    from climate_amd import engine
    from climate_amd.config import synthetic_config
    from climate_amd.model import get_model
    from climate_amd.trainer import HotPathTrainer
    if shared:
        # round-2 stream table: one child per device, whatever the parent
        class _Tab(dict):
            def __contains__(self, k):
                return dict.__contains__(self, k[0])
            def __getitem__(self, k):
                return dict.__getitem__(self, k[0])
            def __setitem__(self, k, v):
                dict.__setitem__(self, k[0], v)
        engine._SideStream._streams = _Tab()
    dev = torch.device("cuda", 0)
    cfg = synthetic_config(base_channels=32, seq_len=6)
    gen = torch.Generator("cpu").manual_seed(1234)
    x = torch.randn(32, 6, 5, 48, 72, generator=gen).to(dev)
    y = torch.randn(32, 2, 48, 72, generator=gen).to(dev)
    losses = {}
    for overlap in (False, True):
        engine.OVERLAP_WGRAD = overlap
        torch.manual_seed(cfg.seed)
        m = get_model(cfg).to(dev)
        tr = HotPathTrainer(m, use_graph=True, distributed=False, micro_batches=2)
        losses[overlap] = [tr.step(x, y).item() for _ in range(3)]
        torch.cuda.synchronize()
    print(f"model shared={shared}: serial {losses[False]} overlapped {losses[True]}")
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(losses[False], losses[True]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "capture_fork_probe.txt"))
    ap.add_argument("--variant", default=None)
    args = ap.parse_args()
    if args.variant:
        import faulthandler
        faulthandler.enable()
        kind, how = args.variant.split("-")
        if kind == "topo":
            topo(how)
        else:
            model(how == "shared")
        sys.exit(0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for v in ("topo-sibling", "topo-own", "topo-shared", "topo-joinorigin", "topo-prealloc", "topo-prefork",
                  "topo-preforkjoin", "model-own"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--variant", v], capture_output=True,
                               text=True, timeout=600)
            f.write(f"==== {v}: exit code {r.returncode}\n{r.stdout[-3000:]}\n---- stderr (tail)\n{r.stderr[-4000:]}\n")
            f.flush()
            print(v, "exit", r.returncode)
