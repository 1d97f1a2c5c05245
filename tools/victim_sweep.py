"""Every launch family of the library as the VICTIM of a co-resident MFMA kernel: victim on the capturing stream, aggressor on
a forked stream, one hipGraph, REPLAYS replays, outputs against a solo run (bit for bit for the launches that are
deterministic on their own, 5e-6 of the output's magnitude for those that add with float atomics).  The counterpart of
tools/aggressor_sweep.py after the round-3 fix (profiles/r03/coresidency/README.md): is anything else schedule dependent?

    python tools/victim_sweep.py [--out gpurun_out/victim_sweep.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402

REPLAYS = 25


def flat(v, out):
    if isinstance(v, torch.Tensor):
        out.append(v)
    elif isinstance(v, (list, tuple)):
        for u in v:
            flat(u, out)
    return out


def beside(victim, aggressor):
    solo = [t.clone() for t in flat(victim(), [])]
    again = [t.clone() for t in flat(victim(), [])]
    own = max([0.0] + [((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(again, solo)])
    aggressor()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            aggressor()
        out = flat(victim(), [])
        main.wait_stream(side)
    worst = 0.0
    for _ in range(REPLAYS):
        g.replay()
        torch.cuda.synchronize()
        worst = max([worst] + [((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(out, solo)])
    return own, worst


def victims():
    torch.manual_seed(3)
    v = {}
    dev = "cuda"
    for (n, c, h, w) in ((96, 32, 48, 72), (96, 64, 24, 36), (96, 128, 12, 18), (96, 256, 6, 9)):
        tag = f"{n}x{c}x{h}x{w}"
        cr = max(c // 8, 1)
        x = torch.randn(n, c, h, w, device=dev)
        dy = torch.randn(n, c, h, w, device=dev)
        gamma = torch.rand(c, device=dev) + 0.5
        beta = torch.randn(c, device=dev) * 0.1
        wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
        wph, winv = ops.pack_conv3x3_h3(wt)
        w1 = torch.randn(cr, c, 1, 1, device=dev) * 0.3
        w2 = torch.randn(c, cr, 1, 1, device=dev) * 0.3
        w7 = torch.randn(1, 2, 7, 7, device=dev) * 0.1
        v[f"conv3x3 fp16x3 (tuned, no reduction split) {tag}"] = lambda x=x, wph=wph, winv=winv, c=c: ops.conv3x3(x, None, c, wph=wph, winv=winv)
        a, st, pooled = ops.gn_silu_fwd(x, gamma, beta, want_pooled=True)
        v[f"gn_silu_fwd {tag}"] = lambda x=x, gamma=gamma, beta=beta: ops.gn_silu_fwd(x, gamma, beta, want_pooled=True)
        dg = torch.zeros(c, device=dev); db = torch.zeros(c, device=dev)
        v[f"gn_silu_bwd {tag} [atomics]"] = lambda x=x, gamma=gamma, beta=beta, st=st, dy=dy, dg=dg, db=db: ops.gn_silu_bwd(x, gamma, beta, st, dy, dg, db)
        res = ops.se_spatial_gate_fwd(a, pooled, w1, w2, w7, pool_out=(h % 2 == 0 and w % 2 == 0))
        v[f"se_spatial_gate_fwd {tag}"] = lambda a=a, pooled=pooled, w1=w1, w2=w2, w7=w7, h=h, w=w: ops.se_spatial_gate_fwd(a, pooled, w1, w2, w7, pool_out=(h % 2 == 0 and w % 2 == 0))
        out0, z, s, fmap, gate = res[:5]
        dw1 = torch.zeros_like(w1); dw2 = torch.zeros_like(w2); dw7 = torch.zeros_like(w7)
        v[f"gates_bwd (4 launches) {tag}"] = (lambda dy=dy, a=a, s=s, z=z, pooled=pooled, gate=gate, fmap=fmap, w1=w1, w2=w2, w7=w7, dw1=dw1, dw2=dw2, dw7=dw7:
                                              ops.gates_bwd(dy, a, s, z, pooled, gate, fmap, w1, w2, w7, dw1, dw2, dw7, defer_se_wgrad=True))
        if ops.block_tail_supported(c, cr, h, w):
            v[f"block_tail_fwd {tag}"] = lambda x=x, gamma=gamma, beta=beta, w1=w1, w2=w2, w7=w7, h=h, w=w: ops.block_tail_fwd(x, gamma, beta, w1, w2, w7, pool_out=(h % 2 == 0 and w % 2 == 0))
            r = ops.block_tail_fwd(x, gamma, beta, w1, w2, w7, pool_out=False)
            y2_, st_, _, z_, s_, fmap_, gate_ = r[:7]
            v[f"block_tail_bwd {tag}"] = (lambda dy=dy, y2_=y2_, st_=st_, gamma=gamma, beta=beta, s_=s_, z_=z_, gate_=gate_, fmap_=fmap_, w1=w1, w2=w2, w7=w7, dw7=dw7:
                                          ops.block_tail_bwd(dy, y2_, st_, gamma, beta, s_, z_, gate_, fmap_, w1, w2, w7, dw7))
        if h % 2 == 0 and w % 2 == 0:
            mp = ops.maxpool2_fwd(x)
            dmp = torch.randn_like(mp)
            v[f"maxpool2_bwd {tag}"] = lambda x=x, dmp=dmp: ops.maxpool2_bwd(x, dmp)
        v[f"time_mean {tag}"] = lambda x=x, n=n: ops.time_mean(x, n // 6, 6)
    # ConvLSTM fused step
    b, ch = 16, 128
    wl = torch.randn(4 * ch, 3 * ch, 3, 3, device=dev) * 0.02
    wph, winv = ops.pack_conv3x3_h3(wl, c_off=2 * ch, cin=ch)
    hp = torch.tanh(torch.randn(b, ch, 6, 9, device=dev)); cprev = torch.randn(b, ch, 6, 9, device=dev)
    gx = torch.randn(b, 4 * ch, 6, 9, device=dev); co = torch.empty_like(cprev); ho = torch.empty_like(cprev)

    gx0 = gx.clone()

    def lstm():
        gx.copy_(gx0)               # (the launch overwrites the pre-activations with the gate activations)
        ops.lstm_step_fwd(hp, wph, winv, gx, cprev, co, ho)
        return [co, ho, gx]
    v["lstm_step_fwd 16x128x6x9"] = lstm
    # transposed conv, head
    xt = torch.randn(16, 128, 12, 18, device=dev); wtt = torch.randn(128, 64, 2, 2, device=dev) * 0.1; bt = torch.randn(64, device=dev)
    v["convT2x2_fwd 16x128x12x18"] = lambda: ops.convT2x2_fwd(xt, wtt, bt)
    xh = torch.randn(16, 32, 48, 72, device=dev); wh = torch.randn(2, 32, 1, 1, device=dev); bh = torch.randn(2, device=dev)
    v["head_fwd 16x32x48x72"] = lambda: ops.head_fwd(xh, wh, bh)
    # cnn_transformer pieces
    M, E = 6912, 256
    t = torch.randn(M, E, device=dev); wq = torch.randn(3 * E, E, device=dev) * 0.05; bq = torch.randn(3 * E, device=dev)
    v["gemm fp16x3 6912x768x256"] = lambda: ops.gemm(t, wq, M, 3 * E, E, bias=bq)
    g_ = torch.rand(E, device=dev) + 0.5; b_ = torch.randn(E, device=dev)
    v["layernorm_fwd 6912x256"] = lambda: ops.layernorm_fwd(t, None, g_, b_)
    qkv = torch.randn(M, 3 * E, device=dev)
    v["attention_fwd 32x216x256 (8 heads)"] = lambda: ops.attention_fwd(qkv, 32, 216, E, 8)
    return v


def aggressors():
    torch.manual_seed(1)
    n = 96
    hp = torch.tanh(torch.randn(n, 128, 6, 9, device="cuda"))
    dA = torch.randn(n, 512, 6, 9, device="cuda")
    gl = torch.zeros(512, 9, 384, device="cuda")
    bex, bey = ops.SampleExponents.measure(hp), ops.SampleExponents.measure(dA)
    x = torch.randn(n, 32, 48, 72, device="cuda")
    wt = torch.randn(32, 32, 3, 3, device="cuda") * 0.02
    wph, winv = ops.pack_conv3x3_h3(wt)
    return {
        "wgrad3x3 fp16x3 cfg 0 (128->512 @6x9)": lambda: ops.wgrad3x3(hp, dA, gl, c_off=256, be_x=bex, be_y=bey, config=ops.H3_BASE + 0 + (4 << 8)),
        "conv3x3 fp16x3 cfg 7 (32->32 @48x72)": lambda: ops.conv3x3(x, None, 32, wph=wph, winv=winv, config=ops.H3_BASE + 7),
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "victim_sweep.txt"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    ops.BLOCK_TAIL_MIN_N = 0
    with open(args.out, "w") as f:
        def say(*a):
            line = " ".join(str(v) for v in a)
            print(line)
            f.write(line + "\n")
            f.flush()
        ag = aggressors()
        bad = 0
        for name, vic in victims().items():
            if args.only and args.only not in name:
                continue
            try:
                vic()
                torch.cuda.synchronize()
            except Exception as e:          # noqa: BLE001
                say(f"{name}: not applicable ({str(e)[:70]})")
                continue
            for an, a in ag.items():
                own, worst = beside(vic, a)
                verdict = "ok" if worst <= max(own * 4, 0.0) or worst <= 5e-6 else "DIFFERS"
                bad += verdict != "ok"
                say(f"{name:58s} beside {an:40s}: solo run-to-run {own:.1e}, worst of {REPLAYS} replays {worst:.1e}  {verdict}")
        say(f"{bad} victim / aggressor pairs differ")
