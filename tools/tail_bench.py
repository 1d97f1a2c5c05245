"""Device time of the sample-resident ConvBlock tail and of the fused ConvLSTM step against the launch chains they replace.

Every candidate is recorded REPS times into a hipGraph and the graph replayed (eager timing of launches shorter than the
host's launch interval measures the host); the figure is microseconds per invocation, launches alone on
the device.

    python tools/tail_bench.py [--out gpurun_out/tail_bench.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402

REPS = 20


def graph_time(fn, reps=REPS, replays=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def tail_case(n, c, h, w, out):
    cr = c // 8
    torch.manual_seed(0)
    y2 = torch.randn(n, c, h, w, device="cuda")
    gamma = torch.ones(c, device="cuda"); beta = torch.zeros(c, device="cuda")
    w1 = torch.randn(cr, c, 1, 1, device="cuda") * 0.3
    w2 = torch.randn(c, cr, 1, 1, device="cuda") * 0.3
    w7 = torch.randn(1, 2, 7, 7, device="cuda") * 0.1
    dout = torch.randn(n, c, h, w, device="cuda")
    pool = h % 2 == 0 and w % 2 == 0

    def chain_fwd():
        a2, st, pooled = ops.gn_silu_fwd(y2, gamma, beta, want_pooled=True)
        return (a2, st, pooled) + tuple(ops.se_spatial_gate_fwd(a2, pooled, w1, w2, w7, pool_out=pool))

    def tail_fwd():
        return ops.block_tail_fwd(y2, gamma, beta, w1, w2, w7, pool_out=pool)

    a2, st, pooled, out0, z, s, fmap, gate = chain_fwd()[:8]
    dw1 = torch.zeros_like(w1); dw2 = torch.zeros_like(w2); dw7 = torch.zeros_like(w7)
    dg = torch.zeros(c, device="cuda"); db = torch.zeros(c, device="cuda")

    def chain_bwd():
        return ops.gates_bwd(dout, a2, s, z, pooled, gate, fmap, w1, w2, w7, dw1, dw2, dw7, defer_se_wgrad=True)

    def tail_bwd():
        return ops.block_tail_bwd(dout, y2, st, gamma, beta, s, z, gate, fmap, w1, w2, w7, dw7)

    dmap, (umax, cnt), dpool, (dsig, dz) = chain_bwd()

    def gn_bwd():
        return ops.gn_silu_bwd_gated(y2, gamma, beta, st, a2, dout, gate, dmap, umax, cnt, s, dpool, dg, db,
                                     se=(dsig, dz, z, pooled, dw1, dw2))

    mb = n * c * h * w * 4 / 1e6
    t = [graph_time(f) for f in (chain_fwd, tail_fwd, chain_bwd, tail_bwd, gn_bwd)]
    out.write(f"tail  N={n:3d} C={c:3d} {h:2d}x{w:2d} ({mb:5.1f} MB/tensor): fwd chain(3) {t[0]:6.1f} us  fused {t[1]:6.1f} us | "
              f"bwd chain(4) {t[2]:6.1f} us  fused {t[3]:6.1f} us | gated GroupNorm bwd {t[4]:6.1f} us\n")
    out.flush()


def lstm_case(b, ch, h, w, out):
    cx = 2 * ch
    torch.manual_seed(1)
    wl = torch.randn(4 * ch, cx + ch, 3, 3, device="cuda") * 0.02
    wph, winv = ops.pack_conv3x3_h3(wl, c_off=cx, cin=ch)
    hp = torch.tanh(torch.randn(b, ch, h, w, device="cuda"))
    cp = torch.randn(b, ch, h, w, device="cuda")
    gx = torch.randn(b, 4 * ch, h, w, device="cuda")
    co = torch.empty_like(cp); ho = torch.empty_like(cp)
    pbuf = [None]

    def chain():
        parts = ops.conv3x3_parts(hp, 4 * ch, wph, winv, parts=pbuf[0])
        pbuf[0] = parts[0]
        ops.lstm_gates_fwd(gx, cp, co, ho, parts=parts)

    def fused():
        ops.lstm_step_fwd(hp, wph, winv, gx, cp, co, ho)

    t = [graph_time(f) for f in (chain, fused)]
    out.write(f"lstm  B={b:3d} Ch={ch:3d} {h}x{w}: projection (partial slices) + gates {t[0]:6.1f} us  fused step {t[1]:6.1f} us\n")
    out.flush()
    if not ops.lstm_step_bwd_supported(b, ch, h, w):
        return
    # backward step: gate backward (with the previous step's partial slices) + recurrent data gradient as partial slices
    wpd, wdinv = ops.pack_conv3x3_h3(wl, c_off=cx, cin=ch, dgrad=True)
    dA_next = torch.randn(b, 4 * ch, h, w, device="cuda")
    act = torch.rand(b, 4 * ch, h, w, device="cuda")
    dc = torch.randn(b, ch, h, w, device="cuda")
    dbuf = [None]

    def chain_b():
        parts = ops.conv3x3_parts(dA_next, ch, wpd, wdinv, parts=dbuf[0])
        dbuf[0] = parts[0]
        ops.lstm_gates_bwd(act, cp, co, None, parts, dc, first=False)

    def fused_b():
        ops.lstm_step_bwd(dA_next, wpd, wdinv, act, cp, co, None, dc)

    t = [graph_time(f) for f in (chain_b, fused_b)]
    out.write(f"lstm  B={b:3d} Ch={ch:3d} {h}x{w}: backward: data gradient (partial slices) + gate backward {t[0]:6.1f} us  "
              f"fused step {t[1]:6.1f} us\n")
    out.flush()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tail_bench.txt"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for case in ((96, 64, 24, 36), (96, 128, 12, 18), (96, 256, 6, 9), (16, 128, 12, 18), (16, 64, 24, 36),
                     (192, 64, 24, 36)):
            tail_case(*case, f)
        for case in ((16, 128, 6, 9), (32, 128, 6, 9), (16, 256, 6, 9)):
            lstm_case(*case, f)
    print(open(args.out).read())
