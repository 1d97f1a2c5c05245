"""Does the sample-resident ConvBlock tail backward (cm_block_tail_bwd) give the same result when another kernel shares the
compute units with it?  tools/overlap_race_probe.py traced the eager / captured side-stream mismatch to this launch: with the
ConvLSTM's h-part weight gradient running on a forked stream, cm_block_tail_bwd of the H/8 level returned a different dmap
for 10-25 % of the samples while umax / cnt stayed right.  This probe isolates the pair: victim on one stream, aggressor on
a forked stream, both recorded into one hipGraph, replayed REPLAYS times; every replay's outputs are compared with a solo run.

Outcome (round 3, profiles/r03/coresidency/README.md): the wrong values sit in lanes 48-63 of whichever wave executes the
tail's 7x7 tap loop, and the instruction responsible is `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (packed fp32 with the
halves of src1 swapped), which the SLP vectorizer had made of the loop's multiplies.  --pk / --canary run the synthetic
detectors of tools/canary (one instruction form, or parked registers / LDS / barriers / loads) in the victim's place;
tools/pk_forms_probe.py characterises the encodings, tools/aggressor_sweep.py the launches that trigger it, and
tools/isa_lint.py keeps the form out of the shipped library.  (The `dgate partials` dump needs a probe build of
cm_block_tail_bwd that exported them; the shipped library does not, and the probe then skips that part.)

    python tools/coresidency_probe.py [--pk | --canary] [--out gpurun_out/coresidency_probe.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402

REPLAYS = 40


def victim_case(n, c, h, w):
    cr = c // 8
    torch.manual_seed(0)
    y2 = torch.randn(n, c, h, w, device="cuda")
    gamma = torch.rand(c, device="cuda") + 0.5
    beta = torch.randn(c, device="cuda") * 0.1
    w1 = torch.randn(cr, c, 1, 1, device="cuda") * 0.3
    w2 = torch.randn(c, cr, 1, 1, device="cuda") * 0.3
    w7 = torch.randn(1, 2, 7, 7, device="cuda") * 0.1
    dout = torch.randn(n, c, h, w, device="cuda")
    pool = h % 2 == 0 and w % 2 == 0
    res = ops.block_tail_fwd(y2, gamma, beta, w1, w2, w7, pool_out=pool)
    y2, st, pooled, z, s, fmap, gate = res[:7]
    dw7 = torch.zeros_like(w7)

    import ctypes
    from climate_amd._lib import lib
    dbg = None
    if hasattr(lib, "cm_block_tail_debug_buffer") and (h, w) == (6, 9):
        lib.cm_block_tail_debug_buffer.argtypes = [ctypes.c_void_p]
        lib.cm_block_tail_debug_buffer.restype = None
        dbg = torch.zeros(n, 3, 32, 64, device="cuda")
        lib.cm_block_tail_debug_buffer(dbg.data_ptr())

    def run():
        dmap, (umax, cnt), dpool, (dsig, dz) = ops.block_tail_bwd(dout, y2, st, gamma, beta, s, z, gate, fmap, w1, w2, w7, dw7)
        r = {"dmap": dmap, "umax": umax, "cnt": cnt, "dpool": dpool, "dsig": dsig, "dz": dz}
        if dbg is not None:
            r["dgate partials as written"] = dbg[:, 0]
            r["dgate partials as read"] = dbg[:, 1]
        return r
    def explain(got, ref, say, name):
        """dmap = conv7^T(dgpre) is linear in dgpre = dgate * g * (1 - g): recover the dgpre the launch must have used from
        its dmap (least squares, 108 equations for 54 unknowns) and show where it differs from the right one."""
        import torch.nn.functional as F
        hw = h * w
        basis = torch.eye(hw, device="cuda", dtype=torch.float64).view(hw, 1, h, w)
        A = F.conv_transpose2d(basis, w7.double(), padding=3).reshape(hw, 2 * hw).t()       # [2*HW, HW]
        if "dgate partials as read" in got:
            wr, rd = got["dgate partials as written"], got["dgate partials as read"]
            wr0 = ref["dgate partials as written"]
            live = torch.zeros(64, dtype=torch.bool, device="cuda")
            live[:54] = True
            dw = (wr - wr0)[:, :, live].abs()
            dr = (rd - wr)[:, :, live].abs()
            say(f"{name}: live dgate partials [sample, slice, 54]: written differently from the solo run: "
                f"{int((dw > 0).sum())} values in {int((dw.amax((1, 2)) > 0).sum())} samples; read differently from what "
                f"the same launch wrote: {int((dr > 0).sum())} values in {int((dr.amax((1, 2)) > 0).sum())} samples")
            if (dw > 0).any():
                smp = int(dw.amax((1, 2)).argmax())
                sl = (dw[smp].amax(1) > 0).nonzero().flatten().tolist()
                say(f"      sample {smp}: slices with wrong written values {sl}; per slice: wrong lanes "
                    f"{[(k, (dw[smp, k] > 0).nonzero().flatten().tolist()) for k in sl[:6]]}")
                k = sl[0]
                say(f"      slice {k}: written {wr[smp, k, :8].tolist()}")
                say(f"      slice {k}: solo    {wr0[smp, k, :8].tolist()}")
        d = (got["dmap"] - ref["dmap"]).abs().reshape(n, -1).amax(1)
        for smp in d.topk(4).indices.tolist():
            for ch in range(2):
                gg, rr = got["dmap"][smp, ch].flatten(), ref["dmap"][smp, ch].flatten()
                bad = ((gg - rr).abs() > 1e-6 * rr.abs().max()).nonzero().flatten().tolist()
                say(f"{name}: sample {smp} dmap channel {ch}: pixels that differ: {bad}")
                if bad:
                    say("      right: " + " ".join(f"{v:+.4f}" for v in rr.tolist()))
                    say("      got  : " + " ".join(f"{v:+.4f}" for v in gg.tolist()))
                    # is it another sample's plane?
                    oth = (ref["dmap"][:, ch].reshape(n, -1) - gg[None]).abs().amax(1)
                    say(f"      closest right plane of any sample: sample {int(oth.argmin())} (max diff {oth.min().item():.2e})")
        for smp in d.topk(0).indices.tolist():
            # hypothesis A: the right w7 and some other dgpre; hypothesis B: the right dgpre and some other w7
            bvec = got["dmap"][smp].double().reshape(-1, 1)
            solA = torch.linalg.lstsq(A, bvec).solution
            resA = (A @ solA - bvec).norm() / bvec.norm()
            e_ref_ = torch.linalg.lstsq(A, ref["dmap"][smp].double().reshape(-1, 1)).solution.view(1, 1, h, w)
            wb = torch.eye(98, device="cuda", dtype=torch.float64).view(98, 1, 2, 7, 7)
            Bm = torch.stack([F.conv_transpose2d(e_ref_, wb[i], padding=3).reshape(-1) for i in range(98)], 1)   # [2*HW, 98]
            solB = torch.linalg.lstsq(Bm, bvec).solution
            resB = (Bm @ solB - bvec).norm() / bvec.norm()
            dw = (solB.view(-1) - w7.double().view(-1)).abs()
            say(f"{name}: sample {smp}: residual of 'right w7, other dgpre' {resA.item():.1e}; of 'right dgpre, other w7' "
                f"{resB.item():.1e} (taps off by > 1e-4: {(dw > 1e-4).nonzero().flatten().tolist()})")
            say(f"      w7 right : {[round(v, 4) for v in w7.view(-1)[:14].tolist()]}")
            say(f"      w7 fitted: {[round(v, 4) for v in solB.view(-1)[:14].tolist()]}")
            e_got = torch.linalg.lstsq(A, got["dmap"][smp].double().reshape(-1, 1)).solution.view(h, w)
            e_ref = torch.linalg.lstsq(A, ref["dmap"][smp].double().reshape(-1, 1)).solution.view(h, w)
            rel = ((e_got - e_ref) / e_ref.abs().max()).cpu()
            say(f"{name}: sample {smp}: (recovered dgpre - right dgpre) / max|dgpre| per pixel:")
            for row in rel.tolist():
                say("      " + " ".join(f"{v:+8.1e}" if abs(v) > 1e-5 else "       0" for v in row))
            ratio = (e_got / e_ref).cpu()
            say("      ratio got / right: " + " ".join(f"{v:.3f}" for v in ratio.flatten().tolist()))
    run.explain = explain
    run.inputs = {"dout": dout, "y2": y2, "stats": st, "s": s, "z": z, "gate": gate, "fmap": fmap, "w7": w7, "gamma": gamma}
    return run


def aggressors(n):
    torch.manual_seed(1)
    hp = torch.tanh(torch.randn(n, 128, 6, 9, device="cuda"))
    dA = torch.randn(n, 512, 6, 9, device="cuda")
    gl = torch.zeros(512, 9, 384, device="cuda")
    bex, bey = ops.SampleExponents.measure(hp), ops.SampleExponents.measure(dA)
    big = torch.ones(1 << 25, device="cuda")
    x = torch.randn(n, 256, 6, 9, device="cuda")
    wt = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
    wph, winv = ops.pack_conv3x3_h3(wt)
    ops.wgrad3x3(hp, dA, gl, c_off=256, be_x=bex, be_y=bey)
    tuned = ops._TUNED.get(("wgrad3x3", n, 6, 9, 128, 0, 512, ops._WG_NUM)) if hasattr(ops, "_TUNED") else None
    variants = {}
    for c in (0,):
        variants[f"wgrad3x3 fp16x3 configuration {c}, 4 quarter rounds"] = (
            lambda c=c: ops.wgrad3x3(hp, dA, gl, c_off=256, be_x=bex, be_y=bey, config=ops.H3_BASE + c + (4 << 8)))
    variants["wgrad3x3 fp32-MFMA configuration 0"] = lambda: ops.wgrad3x3(hp, dA, gl, c_off=256, config=0 + (4 << 8))
    return {
        f"wgrad3x3 (ConvLSTM h-part shape, tuned configuration {tuned})": lambda: ops.wgrad3x3(hp, dA, gl, c_off=256, be_x=bex, be_y=bey),
        **variants,
        "elementwise (128 MB)": lambda: big.mul_(1.0001),
        "conv3x3 fp16x3 256->256 @6x9": lambda: ops.conv3x3(x, None, 256, wph=wph, winv=winv),
    }


def canary_lib():
    """tools/canary/libcanary.so, built on first use (no SLP vectorizer: the canary's scalar reference arithmetic must stay
    scalar)."""
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tools", "canary", "canary.hip")
    so = os.path.join(ROOT, "tools", "canary", "libcanary.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
                               "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


def canary_case(nreg, threads, lds_words, blocks=96, spins=40):
    """Victim = tools/canary/canary.hip: parks known values in VGPRs / LDS / behind barriers, re-reads known global data."""
    lib = canary_lib()
    lib.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    lib.canary_fill.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    nsrc = 1 << 20
    src = torch.empty(nsrc, device="cuda", dtype=torch.int32)
    assert lib.canary_fill(src.data_ptr(), nsrc, torch.cuda.current_stream().cuda_stream) == 0
    report = torch.zeros(8, device="cuda", dtype=torch.int32)

    def run():
        report.zero_()
        rc = lib.canary_launch(report.data_ptr(), src.data_ptr(), nsrc, blocks, threads, nreg, spins, lds_words,
                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return {"report": report}
    run.inputs = {"src": src}
    return run


PK_FORMS = {0: "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", 1: "v_pk_mul_f32 op_sel:[1,0]",
            2: "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]", 3: "v_pk_fma_f32 op_sel:[1,0,0]",
            4: "v_pk_mul_f32 (no swizzle)", 5: "v_pk_fma_f32 op_sel_hi:[0,1,1]", 6: "v_mul_f32 x2 (control)"}


def pk_canary_case(form, threads=1024, lds_words=7700, blocks=96, iters=3000):
    import ctypes
    lib = canary_lib()
    lib.pk_canary_launch.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    report = torch.zeros(48, device="cuda", dtype=torch.int32)

    def run():
        report.zero_()
        rc = lib.pk_canary_launch(report.data_ptr(), form, blocks, threads, iters, lds_words,
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return {"report": report}
    run.inputs = {}
    run.pk = True
    return run


def paired_canary(victim, aggressor, say, name):
    if aggressor is not None:
        aggressor()
    victim()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if aggressor is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                aggressor()
        out = victim()
        if aggressor is not None:
            main.wait_stream(side)
    tot = torch.zeros(8, dtype=torch.int64)
    tid_note = ""
    hit = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPLAYS):
        g.replay()
        torch.cuda.synchronize()
        r = out["report"].cpu().to(torch.int64)
        tot += r[:8]
        hit += int(r[:4].sum() > 0)
        if r.numel() >= 44 and int(r[6]) > 0 and not tid_note:
            tid_note = (f"; threadIdx.x changed in {int(r[6])} lanes (by row {r[8:12].tolist()}), e.g. now "
                        f"{r[12:28].tolist()} where it must be {r[28:44].tolist()}")
    e1.record()
    e1.synchronize()
    if getattr(victim, "pk", False):
        say(f"{name}: replays with any mismatch {hit} of {REPLAYS}; wrong results by 16-lane row of the wave: "
            f"{[int(v) for v in tot[:4]]}; workgroups run {int(tot[4])}{tid_note}")
        return
    say(f"{name}: replays with any mismatch {hit} of {REPLAYS}; totals: registers {int(tot[0])}, LDS words {int(tot[1])}, "
        f"barrier phases {int(tot[2])}, global loads {int(tot[3])}; workgroups run {int(tot[4])}")


def paired(victim, aggressor, say, name):
    solo = {k: v.clone() for k, v in victim().items()}
    before = {k: v.clone() for k, v in victim.inputs.items()}
    if aggressor is not None:
        try:
            aggressor()                 # (autotune outside the capture)
            torch.cuda.synchronize()
        except RuntimeError as e:
            say(f"{name}: not applicable ({str(e)[:60]})")
            return
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if aggressor is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                aggressor()
        out = victim()
        if aggressor is not None:
            main.wait_stream(side)
    bad = {k: 0 for k in solo}
    rows = {k: set() for k in solo}
    worst = {k: 0.0 for k in solo}
    first_bad = None
    for _ in range(REPLAYS):
        g.replay()
        torch.cuda.synchronize()
        for k, ref in solo.items():
            d = (out[k] - ref).abs()
            rel = (d.max() / ref.abs().max().clamp_min(1e-30)).item()
            if rel > 1e-6:
                if first_bad is None:
                    first_bad = {kk: vv.clone() for kk, vv in out.items()}
                bad[k] += 1
                worst[k] = max(worst[k], rel)
                rows[k].update((d.reshape(d.shape[0], -1).amax(1) > 1e-6 * ref.abs().max()).nonzero().flatten().tolist())
    if bad["dmap"] and hasattr(victim, "explain"):
        victim.explain(first_bad, solo, say, name)
    touched = [k for k, v in victim.inputs.items() if not torch.equal(v, before[k])]
    again = victim()
    torch.cuda.synchronize()
    resolo = [k for k, v in again.items() if not torch.equal(v, solo[k])]
    say(f"{name}: inputs changed: {touched}; a solo run afterwards differs from the first in: {resolo}")
    say(f"{name}: replays with a mismatch (of {REPLAYS}) " +
        ", ".join(f"{k} {bad[k]} (worst {worst[k]:.1e}, {len(rows[k])} samples ever hit)" for k in solo))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "coresidency_probe.txt"))
    ap.add_argument("--pk", action="store_true", help="pair the aggressors with the packed-FP32 instruction canary only")
    ap.add_argument("--canary", action="store_true", help="also pair the aggressors with the canary kernel")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        def say(*a):
            line = " ".join(str(v) for v in a)
            print(line)
            f.write(line + "\n")
            f.flush()
        ag = aggressors(96)
        if args.pk:
            for form, label in PK_FORMS.items():
                for thr in (1024, 256):
                    v = pk_canary_case(form, threads=thr)
                    paired_canary(v, None, say, f"{label}, {thr}-thread workgroups, alone")
                    for name, a in ag.items():
                        if "configuration 0" in name or "elementwise" in name or "conv3x3" in name:
                            paired_canary(v, a, say, f"{label}, {thr}-thread workgroups, beside {name}")
            sys.exit(0)
        if args.canary:
            for nreg, thr, ldsw in ((28, 1024, 7700), (24, 1024, 7700), (28, 1024, 16000), (28, 256, 7700), (56, 512, 7700)):
                v = canary_case(nreg, thr, ldsw)
                label = f"canary ({nreg} parked registers, {thr} threads, {ldsw * 4 // 1024} KB LDS)"
                paired_canary(v, None, say, label + " alone")
                for name, a in ag.items():
                    paired_canary(v, a, say, label + " beside " + name)
        for shape in ((96, 256, 6, 9),):
            v = victim_case(*shape)
            paired(v, None, say, f"tail bwd {shape} alone")
            for name, a in ag.items():
                paired(v, a, say, f"tail bwd {shape} beside {name}")
