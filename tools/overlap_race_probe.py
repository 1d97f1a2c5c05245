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


_ORIG_RUN = engine._SideStream.run


ORIGIN_ONLY = [False]      # offload only in the half that runs on the capture's origin stream (mid-capture joins are legal there)


def restrict_offload(which):
    """Offload only a subset of the side-stream jobs of each half's backward (call order: 6 decoder weight gradients, the
    ConvLSTM's x-part and h-part, 8 encoder ones) -- the rest runs inline on the parent stream."""
    lo, hi = {"all": (0, 99), "decoder": (0, 6), "lstm.x": (6, 7), "lstm.h": (7, 8), "encoder": (8, 99)}[which]

    def run(self, fn, *tensors):
        k = getattr(self, "_k", 0)
        self._k = k + 1
        o = engine._SideStream.origin
        if self.enabled and (not (lo <= k < hi) or (ORIGIN_ONLY[0] and o is not None and self.main.cuda_stream != o.cuda_stream)):
            fn()
            return
        _ORIG_RUN(self, fn, *tensors)
    engine._SideStream.run = run


_EMPTY, _EMPTY_LIKE = torch.empty, torch.empty_like


def poison_allocations(value):
    """torch.empty / empty_like return float32 device tensors pre-filled with VALUE (None restores): a kernel that reads an
    output buffer before writing it shows up as a changed gradient in a SERIAL step."""
    if value is None:
        torch.empty, torch.empty_like = _EMPTY, _EMPTY_LIKE
        return

    def empty(*a, **k):
        t = _EMPTY(*a, **k)
        return t.fill_(value) if t.is_cuda and t.dtype == torch.float32 else t

    def empty_like(*a, **k):
        t = _EMPTY_LIKE(*a, **k)
        return t.fill_(value) if t.is_cuda and t.dtype == torch.float32 else t
    torch.empty, torch.empty_like = empty, empty_like


TRACE = {}


def trace_backward():
    """Snapshot (clone on the issuing stream) what the backward's launches return, in host call order: where does the first
    difference between a serial and an overlapped step appear?"""
    from climate_amd import ops

    def flat(v, out):
        if isinstance(v, torch.Tensor):
            out.append(v)
        elif isinstance(v, (list, tuple)):
            for u in v:
                flat(u, out)
        return out

    def wrap(mod, name):
        orig = getattr(mod, name)

        def f(*a, **k):
            r = orig(*a, **k)
            idx = sum(1 for key in TRACE if key.startswith(name + "#"))
            TRACE[f"{name}#{idx:02d}"] = [t.clone() for t in flat(r if r is not None else a, [])]
            return r
        setattr(mod, name, f)
    for name in ("block_tail_bwd", "gn_silu_bwd_gated", "gn_silu_bwd", "maxpool2_bwd", "lstm_gates_bwd", "channel_sum"):
        wrap(ops, name)
    for name in ("convlstm_bwd", "_block_bwd", "up_bwd"):
        wrap(engine, name)


JOIN = {"after": None, "left": -1, "names": []}


def join_after(k):
    """With only the ConvLSTM h-part weight gradient offloaded: the parent stream waits for the side stream right before its
    k-th launch wrapper call after the fork (k = 0: no concurrency at all).  The smallest k that shows the mismatch names the
    parent-stream launch that must not run beside the side job."""
    from climate_amd import ops
    restrict_offload("lstm.h")
    inner = engine._SideStream.run
    JOIN["after"] = k

    def run(self, fn, *tensors):
        before = self._k if hasattr(self, "_k") else 0
        inner(self, fn, *tensors)
        o = engine._SideStream.origin
        if self.enabled and before == 7 and not (ORIGIN_ONLY[0] and o is not None and self.main.cuda_stream != o.cuda_stream):
            JOIN["left"], JOIN["side"], JOIN["names"] = JOIN["after"], self.side, []
    engine._SideStream.run = run
    if JOIN.get("wrapped"):
        return
    JOIN["wrapped"] = True

    def wrap(name):
        orig = getattr(ops, name)

        def f(*a, **kw):
            if JOIN["left"] >= 0:
                if JOIN["left"] == 0:
                    torch.cuda.current_stream().wait_stream(JOIN["side"])
                    JOIN["names"].append("<join>")
                JOIN["names"].append(name)
                JOIN["left"] -= 1
            return orig(*a, **kw)
        setattr(ops, name, f)
    for name in ("channel_sum", "block_tail_bwd", "gn_silu_bwd_gated", "conv3x3", "conv3x3_parts", "wgrad3x3", "gn_silu_bwd",
                 "maxpool2_bwd"):
        wrap(name)


def side_job(mode):
    """What runs on the side stream in the h-part slot (everything else inline on the parent):
    zeros   = only a torch.zeros allocation + fill (the real job inline);  measure = only the exponent measurement of hprev;
    wgrad   = the real job, but its exponent table measured on the PARENT stream first (the side runs the one wgrad launch);
    other   = an unrelated elementwise kernel on private memory."""
    from climate_amd import ops
    scratch = {}

    def run(self, fn, *tensors):
        k = getattr(self, "_k", 0)
        self._k = k + 1
        if not self.enabled or k != 7:
            fn()
            return
        if mode == "wgrad":
            orig = ops.SampleExponents.measure
            parent, side = self.main, self.side

            def measure_on_parent(x):
                with torch.cuda.stream(parent):
                    se = orig(x)
                side.wait_stream(parent)
                return se
            ops.SampleExponents.measure = staticmethod(measure_on_parent)
            try:
                _ORIG_RUN(self, fn, *tensors)
            finally:
                ops.SampleExponents.measure = staticmethod(orig)
            return
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            if mode == "zeros":
                scratch["z"] = torch.zeros(192, device="cuda", dtype=torch.int32)
            elif mode == "measure":
                h = tensors[0]
                scratch["m"] = ops.SampleExponents.measure(h.view(-1, *h.shape[2:]))
            elif mode == "other":
                if "o" not in scratch:
                    scratch["o"] = torch.ones(1 << 26, device="cuda")
                scratch["o"].mul_(1.0001)
        fn()
    engine._SideStream.run = run


GRAPH = [False]


def one_step(overlap, x, y, cfg, micro):
    TRACE.clear()
    JOIN["left"] = -1
    engine.OVERLAP_WGRAD = overlap
    torch.manual_seed(cfg.seed)
    m = get_model(cfg).cuda()
    tr = HotPathTrainer(m, lr=0.0, use_graph=False, distributed=False, micro_batches=micro)
    tr.keep_saved = True
    if GRAPH[0]:
        # the same launches recorded into a hipGraph and replayed: concurrency between the streams is then decided on the
        # device, not by the host's launch timing
        tr._fwd_bwd(x, y)          # (plans / pack tables are built on first use, outside the capture)
        torch.cuda.synchronize()
        TRACE.clear()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tr._fwd_bwd(x, y)
        graph.replay()
    else:
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
    for k, v in TRACE.items():
        grads.update({f"trace:{k}.{j}": t.float() for j, t in enumerate(v)})
    return tr.loss.item(), fwd, grads


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--micro", type=int, default=2)
    ap.add_argument("--graph", action="store_true", help="the schedules recorded into a hipGraph and replayed")
    ap.add_argument("--jobs", action="store_true", help="bisect: which part of the h-part job must run on the side stream")
    ap.add_argument("--join", action="store_true", help="bisect: where may the parent stream join the h-part job")
    ap.add_argument("--trace", action="store_true", help="also compare the backward's intermediate tensors")
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
        if args.trace:
            trace_backward()
        l0, f0, g0 = one_step(False, x, y, cfg, args.micro)
        big = max(v.norm().item() for k, v in g0.items() if not k.startswith("trace:"))
        whats = ("serial", "overlap:all", "overlap:lstm.h", "serial+poison:nan", "serial+poison:12345", "overlap:lstm.x",
                 "overlap:decoder", "overlap:encoder")
        if args.trace:
            whats = ("serial", "overlap:lstm.h", "overlap:all")
        if args.graph:
            whats = ("serial", "graph:serial", "graph:all") + tuple(f"graph:join{k}" for k in range(0, 9))
        if args.graph and args.trace:
            whats = ("graph:serial", "graph:join2", "graph:join3")
        if args.jobs:
            whats = ("serial", "overlap:lstm.h") + tuple(f"overlap:job:{m}" for m in ("zeros", "measure", "wgrad", "other"))
        if args.join:
            whats = ("serial",) + tuple(f"overlap:join{k}" for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 14))
        for what in whats:
            poison_allocations(None)
            GRAPH[0] = what.startswith("graph:")
            ORIGIN_ONLY[0] = what.startswith("graph:join")
            if what.startswith("graph:"):
                what_ = what[6:]
                if what_.startswith("join"):
                    join_after(int(what_[4:]))
                elif what_ != "serial":
                    restrict_offload(what_)
            if what.startswith("serial+poison"):
                poison_allocations(float(what.split(":")[1]))
            elif what.startswith("overlap:job:"):
                side_job(what.split(":")[2])
            elif what.startswith("overlap:join"):
                join_after(int(what[12:]))
            elif what.startswith("overlap:"):
                restrict_offload(what.split(":")[1])
            for rep in range(args.reps):
                l1, f1, g1 = one_step(what.startswith("overlap") or (what.startswith("graph:") and what != "graph:serial"), x, y, cfg, args.micro)
                bad_f = [k for k in f0 if not torch.equal(f0[k], f1[k])]
                errs = sorted(((g1[k] - g0[k]).norm().item() / max(g0[k].norm().item(), 1e-3 * big), k) for k in g0
                              if not k.startswith("trace:"))[::-1]
                worst = ", ".join(f"{k} {e:.1e}" for e, k in errs[:4])
                if errs[0][0] > 1e-5 and not errs[0][1].startswith("trace:"):
                    k = errs[0][1]
                    d = (g1[k] - g0[k]).abs()
                    hot = (d > 1e-4 * g0[k].abs().max()).nonzero()
                    box = [(int(hot[:, j].min()), int(hot[:, j].max())) for j in range(hot.shape[1])] if len(hot) else []
                    say(f"{what:16s} rep {rep}: {k} {tuple(g0[k].shape)}: {len(hot)} elements off by > 1e-4 of max|g|, index "
                        f"ranges {box}, max |diff| / max|g| = {(d.max() / g0[k].abs().max()).item():.2e}; distinct values of "
                        f"dim 0: {hot[:, 0].unique().tolist()[:40]}; of dim 1: {hot[:, 1].unique().tolist()[:40] if hot.shape[1] > 1 else []}")
                if args.join and rep == 0:
                    say(f"{what:16s} parent-stream calls after the fork: {JOIN['names']}")
                if args.trace:
                    for k in sorted(k for k in g0 if k.startswith("trace:")):
                        d = (g1[k] - g0[k]).abs()
                        if d.max().item() > 1e-5 * max(g0[k].abs().max().item(), 1e-30):
                            rows = d.reshape(d.shape[0], -1).amax(1)
                            hot = (rows > 1e-5 * g0[k].abs().max()).nonzero().flatten().tolist()
                            say(f"{what:16s} rep {rep}: {k[6:]} {tuple(g0[k].shape)} max|diff|/max|ref| = "
                                f"{(d.max() / g0[k].abs().max()).item():.1e}, rows (dim 0) affected: {hot[:24]}")
                say(f"{what:16s} rep {rep}: loss diff {abs(l1 - l0) / abs(l0):.1e}; forward tensors differing: {len(bad_f)} "
                    f"{bad_f[:3]}; worst gradients: {worst}")
