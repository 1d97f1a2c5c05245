"""Whole-model parity on the MI355X: forward, loss, all 73 gradients, d(input), Adam steps -- against the golden
fixtures captured from the reference and against the CPU oracle on seeded inputs; plus size-independent properties
at BASELINE config 2's full size.  Tolerance: relative L2 <= 1e-4 (north_star); typical observed 1e-6..1e-5."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import climate_amd
    from climate_amd import model, lightning_module, trainer, optim
    return climate_amd


def _make(amd, in_ch, out_ch, base, T, salt=0):
    from climate_amd.model import AttUNetConvLSTM
    m = AttUNetConvLSTM(in_ch, out_ch, base, T)
    m.load_state_dict(oracle.closed_form_params(in_ch, out_ch, base, salt=salt))
    return m.cuda()


def _sub(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize("name", ["model_tiny.npz", "model_tiny_b16.npz"])
def test_forward_backward_vs_reference_fixture(amd, name):
    g = load_golden(name)
    in_ch, out_ch, base, T = (int(v) for v in g["cfg"][:4])
    m = _make(amd, in_ch, out_ch, base, T, salt=int(g["salt"]) if "salt" in g else 0)
    x = g["x"].cuda().requires_grad_()
    pred = m(x)
    assert rel_l2(pred, g["pred"]) < TOL
    loss = F.mse_loss(pred, g["y"].cuda())
    assert abs(loss.item() - float(g["loss1"])) < 1e-5 * abs(float(g["loss1"]))
    loss.backward()
    assert rel_l2(x.grad, g["dx"]) < TOL
    named = dict(m.named_parameters())
    worst = 0.0
    for k, want in _sub(g, "g.").items():
        e = rel_l2(named[k].grad, want)
        worst = max(worst, e)
        assert e < TOL, (k, e)
    assert named["post_conv.0.weight"].grad is None and named["post_conv.0.bias"].grad is None
    print(f"{name}: worst grad rel-L2 {worst:.2e}")


def test_lightning_module_three_adam_steps(amd):
    """training_step -> loss.backward() -> HipAdam.step(), three times, vs the reference's parameters after 1 and 3
    torch.optim.Adam steps (this is Lightning's automatic optimisation, spelled out)."""
    from climate_amd.lightning_module import ClimateEmulationModule
    g = load_golden("model_tiny.npz")
    in_ch, out_ch, base, T = (int(v) for v in g["cfg"][:4])
    lm = ClimateEmulationModule(_make(amd, in_ch, out_ch, base, T), learning_rate=5e-4, weight_decay=0.0)
    opt = lm.configure_optimizers()
    batch = (g["x"].cuda(), g["y"].cuda())
    for step in range(1, 4):
        opt.zero_grad()
        loss = lm.training_step(batch, 0)
        assert abs(loss.item() - float(g[f"loss{step}"])) < 2e-5 * abs(float(g[f"loss{step}"])), step
        loss.backward()
        opt.step()
        if step in (1, 3):
            sd = lm.model.state_dict()
            for k, want in _sub(g, f"p{step}.").items():
                assert rel_l2(sd[k], want) < 1e-5, (step, k)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_trainer_matches_reference(amd, use_graph):
    from climate_amd.trainer import HotPathTrainer
    g = load_golden("model_tiny.npz")
    in_ch, out_ch, base, T = (int(v) for v in g["cfg"][:4])
    m = _make(amd, in_ch, out_ch, base, T)
    tr = HotPathTrainer(m, lr=5e-4, weight_decay=0.0, use_graph=use_graph)
    x, y = g["x"].cuda(), g["y"].cuda()
    for step in range(1, 4):
        loss = tr.step(x, y)
        assert abs(loss.item() - float(g[f"loss{step}"])) < 2e-5 * abs(float(g[f"loss{step}"])), step
        if step in (1, 3):
            sd = m.state_dict()
            for k, want in _sub(g, f"p{step}.").items():
                assert rel_l2(sd[k], want) < 1e-5, (step, k)
    # post_conv is carried in the state_dict but never updated
    assert torch.equal(m.state_dict()["post_conv.0.weight"].cpu(), g["p1.post_conv.0.weight"])


def test_oracle_parity_seeded_medium(amd):
    """Seeded random input at a mid size (base 16, T=4, B=3, 24x40): HIP vs the CPU oracle run here."""
    in_ch, out_ch, base, T, B, H, W = 5, 2, 16, 4, 3, 24, 40
    P = oracle.closed_form_params(in_ch, out_ch, base, salt=5)
    gen = torch.Generator("cpu").manual_seed(99)
    x = torch.randn(B, T, in_ch, H, W, generator=gen); y = torch.randn(B, out_ch, H, W, generator=gen)
    x[1, :2] = 0.0
    pc = {k: v.clone().requires_grad_() for k, v in P.items()}
    lc = oracle.training_loss(pc, x, y); lc.backward()
    m = _make(amd, in_ch, out_ch, base, T, salt=5)
    pred = m(x.cuda()); lg = F.mse_loss(pred, y.cuda()); lg.backward()
    assert abs(lg.item() - lc.item()) < 1e-5 * lc.item()
    named = dict(m.named_parameters())
    for k in pc:
        if pc[k].grad is not None:
            assert rel_l2(named[k].grad, pc[k].grad) < TOL, k


def test_cfg2_full_size_checksums(amd):
    """BASELINE config 2 (B=32, T=6, base 32, 48x72) through the autograd bridge (what pl.Trainer drives): loss, output
    and every gradient norm vs the reference's own run."""
    g = load_golden("cfg2_checksums.npz")
    in_ch, out_ch, base, T, B, H, W = (int(v) for v in g["cfg"])
    m = _make(amd, in_ch, out_ch, base, T)
    gen = torch.Generator("cpu").manual_seed(int(g["seed"]))
    x = torch.randn(B, T, in_ch, H, W, generator=gen); y = torch.randn(B, out_ch, H, W, generator=gen)
    pred = m(x.cuda()); loss = F.mse_loss(pred, y.cuda()); loss.backward()
    _check_checksums(g, loss.item(), {k: p.grad for k, p in m.named_parameters() if p.grad is not None}, pred)


def test_properties_full_size(amd):
    """Size-independent properties at config-2 size: per-sample independence (a sub-batch / a permuted batch gives
    the same rows; only the split-K convolutions' atomic summation order may differ, hence 1e-6 instead of bit
    equality), eval/no_grad forward equals the training forward, strided input is accepted."""
    in_ch, out_ch, base, T, B, H, W = 5, 2, 32, 6, 32, 48, 72
    m = _make(amd, in_ch, out_ch, base, T)
    gen = torch.Generator("cpu").manual_seed(5)
    x = torch.randn(B, T, in_ch, H, W, generator=gen).cuda()
    with torch.no_grad():
        full = m(x)
        half = m(x[:16])
        m.eval()
        ev = m(x[:16])
    assert torch.isfinite(full).all()
    assert rel_l2(half, full[:16]) < 1e-6 and rel_l2(ev, half) < 1e-6
    perm = torch.randperm(B, generator=gen)
    with torch.no_grad():
        pp = m(x[perm.cuda()])
    assert rel_l2(pp, full[perm.cuda()]) < 1e-6
    # non-contiguous input (a strided view) is accepted, like nn.Conv2d accepts it
    xt = x[:4].transpose(0, 1).contiguous().transpose(0, 1)
    assert not xt.is_contiguous()
    with torch.no_grad():
        assert rel_l2(m(xt), full[:4]) < 1e-6


def test_error_behaviour(amd):
    m = _make(amd, 5, 2, 8, 3)
    with pytest.raises(RuntimeError, match="channel mismatch"):
        m(torch.zeros(1, 3, 7, 16, 24, device="cuda"))
    with pytest.raises(RuntimeError, match="divisible by 8"):
        m(torch.zeros(1, 3, 5, 12, 24, device="cuda"))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 5, 16, 24, device="cuda"))


def test_checkpoint_roundtrip(amd):
    from climate_amd.model import AttUNetConvLSTM
    m = _make(amd, 5, 2, 8, 3)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    m2 = AttUNetConvLSTM(5, 2, 8, 3)
    m2.load_state_dict(sd)
    m2 = m2.cuda()
    x = torch.randn(2, 3, 5, 16, 24, device="cuda")
    with torch.no_grad():
        assert rel_l2(m(x), m2(x)) < 1e-6


def _se_margin(P, x):
    """Smallest |pre-ReLU SE hidden activation| relative to the largest one, over every SE block of the oracle's
    forward.  A hidden unit sitting at ~0 makes the ReLU mask (hence the gradients) flip between any two fp32
    evaluations, which is a property of the model, not of a kernel; such parameter sets are not used for parity."""
    import oracle.cpu_ref as ref
    seen = []
    orig = ref.se_block

    def spy(xx, w1, w2):
        z = F.conv2d(xx.mean(dim=(2, 3), keepdim=True), w1)
        seen.append((z.abs().min() / z.abs().max().clamp_min(1e-30)).item())
        return orig(xx, w1, w2)
    ref.se_block = spy
    try:
        with torch.no_grad():
            oracle.model_forward(P, x)
    finally:
        ref.se_block = orig
    return min(seen)


@pytest.mark.parametrize("shape", [
    dict(base=64, T=6, B=1, H=192, W=288),      # BASELINE config 5: base 64, seq_len 6 on the 192x288 grid
    dict(base=64, T=12, B=4, H=48, W=72),       # BASELINE config 3: base 64, seq_len 12
    dict(base=16, T=2, B=1, H=192, W=288),      # narrow channels on the wide grid (other tile / vector-width choices)
])
def test_baseline_configs_3_and_5_vs_fp64_oracle(amd, shape):
    """BASELINE.json configs 3 and 5 at their DEFINING channel widths, sequence length and grid (batch cut so the
    fp64 oracle finishes in about a minute), every gradient at 1e-4.

    The loss gradient is discontinuous in the CBAM channel-argmax and the MaxPool argmax, and at these sizes some
    pixel always sits on such a discontinuity (top-2 gap ~1e-7 relative: the best of 160 parameter / input draws had a
    minimum amax gap of 2.9e-6, MaxPool windows ~1e-7), where two correct fp32 evaluations may choose differently.  So
    the comparison is decision-aware (oracle.Decisions): the fp64 oracle's backward adopts the device path's choices
    and VALIDATES each one -- the chosen element must lie within 1e-5 * (|max| + rms) of the oracle's own maximum -- i.e.
    the two may differ only where the reference function itself is ambiguous; a wrong choice is a failure
    (``violations == 0``), and everything else is held to 1e-4 with no per-shape tolerance."""
    from climate_amd import engine
    from _decisions import hip_decisions
    in_ch, out_ch = 5, 2
    base, T, B, H, W = shape["base"], shape["T"], shape["B"], shape["H"], shape["W"]
    gen = torch.Generator("cpu").manual_seed(321)
    x = torch.randn(B, T, in_ch, H, W, generator=gen); y = torch.randn(B, out_ch, H, W, generator=gen)
    for salt in range(9, 30):
        P = oracle.closed_form_params(in_ch, out_ch, base, salt=salt)
        if _se_margin(P, x) > 2e-3:           # (a handful of SE hidden units per model: selectable, unlike pixels)
            break
    else:
        pytest.skip("no parameter set with a safe ReLU margin found")
    m = _make(amd, in_ch, out_ch, base, T, salt=salt)
    p = m._param_dict()
    g = m._views(torch.zeros(m.n_flat_trainable, device="cuda"))
    pk = engine.get_plan(p, None, False).pack()
    pred, sv = engine.forward(p, pk, x.cuda(), save=True)
    dec = hip_decisions(sv)
    yd = y.cuda()
    loss_hip = F.mse_loss(pred, yd).item()
    engine.backward(p, pk, g, sv, (2.0 / pred.numel()) * (pred - yd))
    torch.cuda.synchronize()
    pc = {k: v.double().requires_grad_() for k, v in P.items()}
    lc = oracle.training_loss(pc, x.double(), y.double(), decisions=dec); lc.backward()
    print(f"{shape}: salt {salt}; {dec.sites} decision sites, {dec.differing} chosen differently from the oracle's own "
          f"(all within 1e-5 of its maximum), violations {dec.violations}")
    assert dec.violations == 0, dec.log
    assert dec.differing <= 1e-4 * dec.sites            # ambiguity is rare by construction
    assert abs(loss_hip - lc.item()) < 1e-5 * abs(lc.item())
    errs = sorted(((rel_l2(g[k], pc[k].grad), k) for k in pc if pc[k].grad is not None), reverse=True)
    print(f"{shape}: worst grad rel-L2: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:4]))
    import os
    if os.environ.get("CM_TEST_REF32"):
        p32 = {k: v.float().requires_grad_() for k, v in P.items()}
        oracle.training_loss(p32, x, y, decisions=hip_decisions(sv)).backward()
        print("   fp32 CPU oracle vs fp64 on the same tensors: " + ", ".join(f"{k} {rel_l2(p32[k].grad, pc[k].grad):.2e}" for e, k in errs[:6]))
        print("   device on the same tensors: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:6]))
    assert errs[0][0] < TOL, errs[0]
    with torch.no_grad():
        assert rel_l2(pred, oracle.model_forward(P, x)) < TOL


# ----------------------------------------------------------------------------------------------- full-size, fused path
def _check_checksums(g, loss, grads, pred=None, grad_tol=TOL):
    """loss / prediction / per-tensor gradient norms against the reference's fp32 run, all at 1e-4.  (The 8 sampled
    elements per tensor stored in the fixture are not used: individual elements of the reference's OWN fp32 gradients
    are off by more than 1e-4 of the tensor's RMS wherever an argmax decision is ambiguous -- observed 2.7e-4 on
    enc1.body.4.weight -- so element-level parity at this size is checked against the fp64 oracle with the device's
    decisions imposed, see _check_vs_fp64_oracle.)"""
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    if pred is not None:
        assert abs(pred.double().norm().item() - float(g["pred_l2"])) < 1e-5 * float(g["pred_l2"])
        idx = torch.from_numpy(g["pred_sample_idx"])
        assert rel_l2(pred.flatten().cpu()[idx], g["pred_samples"]) < TOL
    if grads is None:
        return
    bad = []
    for name, want in zip(g["grad_names"].tolist(), g["grad_l2"].tolist()):
        got = grads[name]
        e_norm = abs(got.double().norm().item() - want) / max(want, 1e-30)
        if e_norm > grad_tol:
            bad.append((name, f"norm {e_norm:.2e}"))
    assert not bad, bad


def _check_vs_fp64_oracle(P, x, y, sv, loss, grads, what):
    """Every gradient (all elements, relative L2 <= 1e-4) against the fp64 oracle whose backward adopts -- and
    validates -- the device forward's amax / MaxPool choices (oracle.Decisions)."""
    from _decisions import hip_decisions
    dec = hip_decisions(sv)
    pc = {k: v.double().requires_grad_() for k, v in P.items()}
    lc = oracle.training_loss(pc, x.double(), y.double(), decisions=dec); lc.backward()
    print(f"{what}: {dec.sites} decision sites, {dec.differing} differ from the oracle's own, violations {dec.violations}")
    assert dec.violations == 0, dec.log
    assert dec.differing <= 1e-4 * dec.sites
    assert abs(loss - lc.item()) < 1e-5 * abs(lc.item())
    worst = 0.0
    for k in pc:
        if pc[k].grad is not None:
            e = rel_l2(grads[k], pc[k].grad)
            worst = max(worst, e)
            assert e < TOL, (k, e)
    print(f"{what}: worst grad rel-L2 {worst:.2e}")


@pytest.mark.parametrize("use_graph", [True, False])
def test_cfg2_full_size_through_the_benchmarked_trainer(amd, use_graph):
    """The path bench.py times -- HotPathTrainer: batched pack / zero / unpack, fused head+MSE+head-backward, pruned
    pack table, hipGraph replay -- at BASELINE config 2's full size against the reference's checksums: loss and every
    gradient norm + samples after the first step (gradients stay in the flat buffer until the next step zeroes it)."""
    from climate_amd.trainer import HotPathTrainer
    g = load_golden("cfg2_checksums.npz")
    in_ch, out_ch, base, T, B, H, W = (int(v) for v in g["cfg"])
    m = _make(amd, in_ch, out_ch, base, T)
    gen = torch.Generator("cpu").manual_seed(int(g["seed"]))
    x = torch.randn(B, T, in_ch, H, W, generator=gen); y = torch.randn(B, out_ch, H, W, generator=gen)
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    tr = HotPathTrainer(m, lr=5e-4, use_graph=use_graph, distributed=False)
    tr.keep_saved = True
    loss = tr.step(x.cuda(), y.cuda()).item()
    grads = m._views(tr.grad)
    _check_checksums(g, loss, grads)
    if use_graph:
        _check_vs_fp64_oracle(P, x, y, tr.saved, loss, grads, "cfg2, graphed trainer")
        # replaying the captured graph on the same batch: the loss goes down, gradients stay finite
        l2 = tr.step(x.cuda(), y.cuda()).item()
        assert l2 < loss and torch.isfinite(tr.grad).all()


def test_default_init_left_padded_window(amd):
    """The reference's real edge case end to end (main_final.py:76,127-131 + default init, beta = 0): every channel of
    an all-zero frame is exactly 0 after GroupNorm+SiLU => C-way amax ties and all-equal MaxPool windows on WHOLE
    frames.  Forward, loss, d(input) and all 73 gradients vs the reference fixture."""
    from climate_amd.model import AttUNetConvLSTM
    g = load_golden("model_default_init_padded.npz")
    in_ch, out_ch, base, T = (int(v) for v in g["cfg"][:4])
    m = AttUNetConvLSTM(in_ch, out_ch, base, T)
    m.load_state_dict(_sub(g, "p."))
    m = m.cuda()
    x = g["x"].cuda().requires_grad_()
    pred = m(x)
    assert rel_l2(pred, g["pred"]) < TOL
    loss = F.mse_loss(pred, g["y"].cuda())
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    assert rel_l2(x.grad, g["dx"]) < TOL
    named = dict(m.named_parameters())
    for k, want in _sub(g, "g.").items():
        assert torch.isfinite(named[k].grad).all(), k
        assert rel_l2(named[k].grad, want) < TOL, k


def test_default_init_left_padded_window_cfg2_size(amd):
    """Same edge case at BASELINE config 2's full size THROUGH THE GRAPHED TRAINER: seed-42 default init, 1/8 of the
    samples with their first T-1 frames zeroed (SURVEY 8d second input set).

    Forward (loss, prediction) against the reference's own run (fixture).  Gradients against the fp64 oracle with the
    device's amax / MaxPool choices imposed and validated (oracle.Decisions): in this configuration the reference's
    OWN fp32 gradients are off by up to 4.8e-4 from their fp64 values (zero frames have rstd = 1/sqrt(eps) = 316, which
    amplifies rounding noise; measured: enc4.conv.se.fc.0.weight 4.8e-4, enc1.body.0.weight 1.8e-4), so an fp32
    checksum cannot pin them at 1e-4 -- the fp64 oracle can."""
    from climate_amd.model import AttUNetConvLSTM
    from climate_amd.trainer import HotPathTrainer
    g = load_golden("cfg2_default_init_padded_checksums.npz")
    in_ch, out_ch, base, T, B, H, W = (int(v) for v in g["cfg"])
    torch.manual_seed(42)
    m = AttUNetConvLSTM(in_ch, out_ch, base, T)
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    gen = torch.Generator("cpu").manual_seed(int(g["seed"]))
    x = torch.randn(B, T, in_ch, H, W, generator=gen)
    x[::8, :T - 1] = 0.0
    y = torch.randn(B, out_ch, H, W, generator=gen)
    with torch.no_grad():
        pred = m(x.cuda())
    tr = HotPathTrainer(m, lr=5e-4, use_graph=True, distributed=False)
    tr.keep_saved = True                      # the captured forward's activations stay referenced (decision read-out)
    loss = tr.step(x.cuda(), y.cuda()).item()
    assert torch.isfinite(tr.grad).all()
    _check_checksums(g, loss, None, pred)
    _check_vs_fp64_oracle(P, x, y, tr.saved, loss, m._views(tr.grad), "default init, padded, graphed trainer")


def test_plain_unet_vs_reference_fixture(amd):
    """model.type = unet (src/unet.py:72-109) on the HIP engine vs the reference: forward, d(input), all 80 gradients."""
    from climate_amd.model import UNet
    g = load_golden("unet_tiny.npz")
    in_ch, out_ch, base = (int(v) for v in g["cfg"][:3])
    m = UNet(in_ch, out_ch, base)
    m.load_state_dict(oracle.closed_form_params(in_ch, out_ch, base, salt=int(g["salt"]),
                                                shapes=oracle.unet_param_shapes(in_ch, out_ch, base)))
    m = m.cuda()
    x = g["x"].cuda().requires_grad_()
    pred = m(x)
    assert rel_l2(pred, g["pred"]) < TOL
    F.mse_loss(pred, g["y"].cuda()).backward()
    assert rel_l2(x.grad, g["dx"]) < TOL
    named = dict(m.named_parameters())
    for k, want in _sub(g, "g.").items():
        assert rel_l2(named[k].grad, want) < TOL, k


# ----------------------------------------------------------------------------------------------- autograd contract
def test_gradient_accumulation_and_two_calls_in_one_graph(amd):
    """p.grad must not alias the engine's workspace: two micro-batches without zero_grad accumulate, and
    two model calls inside one autograd graph both contribute -- against the oracle's gradients."""
    in_ch, out_ch, base, T, B, H, W = 5, 2, 8, 3, 2, 16, 24
    P = oracle.closed_form_params(in_ch, out_ch, base)
    gen = torch.Generator("cpu").manual_seed(17)
    xs = [torch.randn(B, T, in_ch, H, W, generator=gen) for _ in range(2)]
    ys = [torch.randn(B, out_ch, H, W, generator=gen) for _ in range(2)]
    pc = {k: v.clone().requires_grad_() for k, v in P.items()}
    (oracle.training_loss(pc, xs[0], ys[0]) + 10.0 * oracle.training_loss(pc, xs[1], ys[1])).backward()
    # (a) two backward passes, no zero_grad in between (Lightning accumulate_grad_batches = 2)
    m = _make(amd, in_ch, out_ch, base, T)
    F.mse_loss(m(xs[0].cuda()), ys[0].cuda()).backward()
    (10.0 * F.mse_loss(m(xs[1].cuda()), ys[1].cuda())).backward()
    named = dict(m.named_parameters())
    for k in pc:
        if pc[k].grad is not None:
            assert rel_l2(named[k].grad, pc[k].grad) < TOL, ("accumulate", k)
    # (b) zero_grad(set_to_none=False) keeps the tensors; the next backward must add into zeros, not into stale data
    m.zero_grad(set_to_none=False)
    F.mse_loss(m(xs[0].cuda()), ys[0].cuda()).backward()
    pc0 = {k: v.clone().requires_grad_() for k, v in P.items()}
    oracle.training_loss(pc0, xs[0], ys[0]).backward()
    for k in pc0:
        if pc0[k].grad is not None:
            assert rel_l2(named[k].grad, pc0[k].grad) < TOL, ("zero_grad", k)
    # (c) both calls in ONE graph
    m2 = _make(amd, in_ch, out_ch, base, T)
    (F.mse_loss(m2(xs[0].cuda()), ys[0].cuda()) + 10.0 * F.mse_loss(m2(xs[1].cuda()), ys[1].cuda())).backward()
    named2 = dict(m2.named_parameters())
    for k in pc:
        if pc[k].grad is not None:
            assert rel_l2(named2[k].grad, pc[k].grad) < TOL, ("one graph", k)


def test_trainer_checkpoint_resume(amd):
    """HotPathTrainer.state_dict()/load_state_dict(): a resumed run continues exactly like the uninterrupted one, and
    the optimizer state is torch.optim.Adam's format (loads into torch.optim.Adam)."""
    from climate_amd.trainer import HotPathTrainer
    g = load_golden("model_tiny.npz")
    in_ch, out_ch, base, T = (int(v) for v in g["cfg"][:4])
    x, y = g["x"].cuda(), g["y"].cuda()
    a = HotPathTrainer(_make(amd, in_ch, out_ch, base, T), lr=5e-4, use_graph=False)
    for _ in range(2):
        a.step(x, y)
    ck = a.state_dict()
    assert len(ck["optimizer"]["state"]) == 73 and float(ck["optimizer"]["state"][0]["step"]) == 2.0
    l3 = a.step(x, y).item()
    b = HotPathTrainer(_make(amd, in_ch, out_ch, base, T, salt=4), lr=1e-2, use_graph=True)
    b.load_state_dict(ck)
    assert b.lr == 5e-4
    l3b = b.step(x, y).item()
    assert abs(l3 - l3b) < 1e-6 * abs(l3)
    assert abs(l3 - float(g["loss3"])) < 2e-5 * float(g["loss3"])
    for k, want in _sub(g, "p3.").items():
        assert rel_l2(b.model.state_dict()[k], want) < 1e-5, k
    cpu = torch.optim.Adam([torch.nn.Parameter(v.cpu().clone()) for v in b.model.state_dict().values()], lr=1.0)
    cpu.load_state_dict(b.optimizer_state_dict())
    assert cpu.param_groups[0]["lr"] == 5e-4


# ----------------------------------------------------------------------------------------------- micro-batch overlap
@pytest.mark.parametrize("use_graph", [False, True])
def test_micro_batch_overlap_equals_one_batch(amd, use_graph):
    """HotPathTrainer(micro_batches=2): the batch's halves run on two streams into two gradient buffers that are
    averaged before Adam.  Loss, every gradient and the parameters after three Adam steps equal the one-batch
    schedule's to rounding (summation order), eagerly and through the captured graph; the automatic choice picks two
    micro-batches at BASELINE config 2's workload and one at the larger configurations."""
    from climate_amd.trainer import HotPathTrainer
    in_ch, out_ch, base, T, B, H, W = 5, 2, 16, 3, 6, 16, 24
    gen = torch.Generator("cpu").manual_seed(31)
    x = torch.randn(B, T, in_ch, H, W, generator=gen).cuda(); y = torch.randn(B, out_ch, H, W, generator=gen).cuda()
    x[1, :T - 1] = 0.0                                   # a left-padded window in the first half
    one = HotPathTrainer(_make(amd, in_ch, out_ch, base, T), lr=5e-4, use_graph=use_graph, distributed=False,
                         micro_batches=1)
    two = HotPathTrainer(_make(amd, in_ch, out_ch, base, T), lr=5e-4, use_graph=use_graph, distributed=False,
                         micro_batches=2)
    for step in range(3):
        l1, l2 = one.step(x, y).item(), two.step(x, y).item()
        assert two._parts == 2 and one._parts == 1
        assert abs(l1 - l2) < 2e-6 * abs(l1), (step, l1, l2)
        g1, g2 = one.model._views(one.grad), two.model._views(two.grad)
        for k in g1:
            assert torch.isfinite(g2[k]).all(), k
            # (TOL, not rounding level: gradients that cancel to ~1e-7 of their terms -- GroupNorm-invariant directions --
            # move by several 1e-5 with the order of the float atomics alone, see test_run_twice_determinism)
            assert rel_l2(g2[k], g1[k]) < TOL, (step, k, rel_l2(g2[k], g1[k]))
    for (k, a), (_, b) in zip(one.model.state_dict().items(), two.model.state_dict().items()):
        assert rel_l2(b, a) < 2e-5, k
    with pytest.raises(ValueError):
        HotPathTrainer(_make(amd, in_ch, out_ch, base, T), use_graph=False, distributed=False,
                       micro_batches=2).step(x[:3], y[:3])
    auto = HotPathTrainer(_make(amd, in_ch, out_ch, 32, 6), use_graph=False, distributed=False)
    assert auto._auto_micro(torch.empty(32, 6, 5, 48, 72, device="meta")) == 2        # BASELINE config 2
    assert auto._auto_micro(torch.empty(32, 12, 5, 48, 72, device="meta")) == 2
    auto.model.base = 64
    assert auto._auto_micro(torch.empty(32, 12, 5, 48, 72, device="meta")) == 2       # config 3
    assert auto._auto_micro(torch.empty(16, 6, 5, 192, 288, device="meta")) == 1      # config 5
    auto.model.base = 32
    assert auto._auto_micro(torch.empty(3, 6, 5, 48, 72, device="meta")) == 1         # odd batch


@pytest.mark.parametrize("use_graph", [False, True])
def test_inference_runner_matches_module_forward(amd, use_graph):
    """InferenceRunner (validation_step / test_step: forward only, main_final.py:563-574) == the module's forward, one
    batch and two halves on two streams, eager and through the replayed graph (second call = replay)."""
    from climate_amd.trainer import InferenceRunner
    in_ch, out_ch, base, T, B, H, W = 5, 2, 16, 3, 6, 16, 24
    m = _make(amd, in_ch, out_ch, base, T)
    gen = torch.Generator("cpu").manual_seed(77)
    x = torch.randn(B, T, in_ch, H, W, generator=gen).cuda()
    x2 = torch.randn(B, T, in_ch, H, W, generator=gen).cuda()
    with torch.no_grad():
        want, want2 = m(x).clone(), m(x2).clone()
    for parts in (1, 2, None):
        inf = InferenceRunner(m, use_graph=use_graph, micro_batches=parts)
        assert rel_l2(inf(x), want) < 2e-6, parts
        assert rel_l2(inf(x2), want2) < 2e-6, parts          # (graphed: a replay with new input)
        assert rel_l2(inf(x), want) < 2e-6, parts


# ----------------------------------------------------------------------------------------------- determinism / overlap
def test_run_twice_determinism(amd):
    """Split-K convolutions and all weight gradients accumulate with fp32 atomics, so two runs are not bit-identical;
    they must agree to rounding level (SURVEY section 5 asked for exactly this check), graph replay included."""
    from climate_amd.trainer import HotPathTrainer
    in_ch, out_ch, base, T, B, H, W = 5, 2, 32, 6, 32, 48, 72
    m = _make(amd, in_ch, out_ch, base, T)
    gen = torch.Generator("cpu").manual_seed(8)
    x = torch.randn(B, T, in_ch, H, W, generator=gen).cuda(); y = torch.randn(B, out_ch, H, W, generator=gen).cuda()
    tr = HotPathTrainer(m, lr=0.0, use_graph=False, distributed=False)     # lr = 0: parameters stay put
    runs = []
    for _ in range(3):
        tr._fwd_bwd(x, y)
        runs.append((tr.grad.clone(), tr.loss.item()))
    tr.use_graph = True
    tr.step(x, y)
    runs.append((tr.grad.clone(), tr.loss.item()))
    g0, l0 = runs[0]
    lay = m._build_layout()
    for gi, li in runs[1:]:
        assert abs(li - l0) < 1e-6 * abs(l0)                   # (the loss is summed with float atomics: +-1 ulp)
        for k, (o, n, _s) in lay.items():
            if o + n <= g0.numel():
                assert rel_l2(gi[o:o + n], g0[o:o + n]) < 5e-5, k       # (observed <= 2.5e-6, SE weights up to 1.5e-5: cancelling sums, atomics order)


@pytest.mark.parametrize("use_graph", [False, True])
def test_side_stream_overlap_three_steps(amd, monkeypatch, use_graph):
    """Weight gradients on side streams (engine.OVERLAP_WGRAD, off by default) beside two micro-batches, three steps at the
    benchmark size: EAGER (the configuration that produced non-finite gradients in round 1) and GRAPH CAPTURED -- the
    nested fork that crashed hipStreamEndCapture in round 2 and is recorded since round 3 with the side streams
    pre-forked from the capture's origin and joined into it (engine._SideStream, profiles/r03/capture_fork_probe.txt).
    Finite, and equal to the serial schedule."""
    from climate_amd import engine
    from climate_amd.trainer import HotPathTrainer
    in_ch, out_ch, base, T, B, H, W = 5, 2, 32, 6, 32, 48, 72
    gen = torch.Generator("cpu").manual_seed(7)
    x = torch.randn(B, T, in_ch, H, W, generator=gen).cuda()
    y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()
    res = {}
    for overlap in (False, True):
        monkeypatch.setattr(engine, "OVERLAP_WGRAD", overlap)
        m = _make(amd, in_ch, out_ch, base, T)
        tr = HotPathTrainer(m, lr=1e-3, use_graph=use_graph, distributed=False)
        losses, g1 = [], None
        for _ in range(3):
            losses.append(tr.step(x, y).item())
            assert tr._parts == 2
            assert torch.isfinite(tr.grad).all(), (overlap, len(losses))
            if g1 is None:
                g1 = tr.grad.clone()
        res[overlap] = (losses, g1)
    for a, b in zip(*[res[k][0] for k in (False, True)]):
        assert abs(a - b) < 1e-5 * abs(a)
    # first-step gradients (identical parameters): the two schedules differ by summation order only.  (Parameters
    # after several Adam steps are not compared: Adam's per-element normalisation turns rounding-level gradient
    # differences on near-zero elements into lr-sized steps.)
    lay = m._build_layout()
    ga, gb = res[True][1], res[False][1]
    big = max(gb[o:o + n].norm().item() for k, (o, n, _s) in lay.items() if o + n <= gb.numel())
    for k, (o, n, _s) in lay.items():
        if o + n <= gb.numel():
            # (summation order of the float atomics differs between the schedules.  On this task some tensors' gradients --
            #  SE weights, GroupNorm scales of the deep levels -- are ~1e-8 sums of strongly cancelling ~1e-5 terms: their
            #  own norm is not a meaningful yardstick (2.5e-4 of it observed), the step's gradient scale is)
            err = (ga[o:o + n] - gb[o:o + n]).norm().item() / max(gb[o:o + n].norm().item(), 1e-3 * big)
            # Both ways of issuing it (captured, and eagerly on four streams) agree with the serial schedule at rounding
            # level.  Rounds 1-3 saw the EAGER schedule up to 6e-4 away (round 1 a non-finite step): root-caused in round 3
            # to a packed-fp32 instruction form that misbehaves beside MFMA waves and is now kept out of the library
            # (profiles/r03/coresidency/); the eager bound was 2e-3 until then.
            assert err < 5e-5, (k, err)


@pytest.mark.parametrize("name", ["cfg3_b32_checksums.npz", "cfg5_b16_checksums.npz"])
def test_configs_3_and_5_at_their_per_rank_batch_vs_reference_checksums(amd, name):
    """BASELINE.json configs 3 and 5 at the PER-RANK batch they define (32 x 12 frames at base 64; 16 x 6 frames of
    192x288 at base 64) through the graphed trainer -- the sample-group, grid-size and micro-batch choices of exactly
    those shapes -- against checksums of the reference's own run (tests/golden/gen_golden_big.py): loss (1e-5),
    prediction norm and samples (1e-5 / 1e-4), every per-tensor gradient norm.  The gradient norms are held to 1e-3 here,
    not 1e-4: the reference's run is fp32, and at these sizes its own gradients sit up to 4.8e-4 from their float64
    values wherever an amax / MaxPool decision is ambiguous (3e-4 observed on enc1.body.1.weight at
    192x288) -- element-level parity at 1e-4 for these widths is the decision-aware float64 test above."""
    from climate_amd.trainer import HotPathTrainer, InferenceRunner
    g = load_golden(name)
    in_ch, out_ch, base, T, B, H, W = (int(v) for v in g["cfg"])
    m = _make(amd, in_ch, out_ch, base, T, salt=int(g["salt"]))
    gen = torch.Generator("cpu").manual_seed(int(g["seed"]))
    x = torch.randn(B, T, in_ch, H, W, generator=gen); y = torch.randn(B, out_ch, H, W, generator=gen)
    xd, yd = x.cuda(), y.cuda()
    pred = InferenceRunner(m, use_graph=False)(xd).clone()
    tr = HotPathTrainer(m, lr=5e-4, use_graph=True, distributed=False)
    loss = tr.step(xd, yd).item()
    _check_checksums(g, loss, m._views(tr.grad), pred, grad_tol=1e-3)
    l2 = tr.step(xd, yd).item()                       # graph replay
    assert l2 < loss and torch.isfinite(tr.grad).all()


def test_graph_replay_equals_eager_over_50_steps(amd):
    """BASELINE config 2, two micro-batches on two streams: 50 graph-replayed steps against 50 eager steps of the same
    schedule on a learnable synthetic task.  The two runs execute the same kernels in the same order; they differ only in
    the order of float atomics (weight-gradient staging), so the loss trajectories must stay together (1e-3 relative after
    50 Adam steps, 2e-5 over the first five) and the loss must fall."""
    from climate_amd.config import synthetic_config
    from climate_amd.model import get_model
    from climate_amd.trainer import HotPathTrainer
    cfg = synthetic_config(base_channels=32, seq_len=6)
    gen = torch.Generator("cpu").manual_seed(7)
    x = torch.randn(32, 6, 5, 48, 72, generator=gen).cuda()
    y = (x[:, -1, :2] * 0.5 + x[:, 0, 1:3] * 0.25).contiguous()          # a learnable target
    losses = {}
    for graph in (True, False):
        torch.manual_seed(cfg.seed)
        m = get_model(cfg).cuda()
        tr = HotPathTrainer(m, lr=1e-3, use_graph=graph, distributed=False, micro_batches=2)
        losses[graph] = torch.stack([tr.step(x, y).clone() for _ in range(50)]).flatten().cpu()
        assert tr._parts == 2
    g, e = losses[True], losses[False]
    assert torch.isfinite(g).all() and torch.isfinite(e).all()
    assert ((g[:5] - e[:5]).abs() <= 2e-5 * e[:5].abs()).all(), (g[:5], e[:5])
    assert ((g - e).abs() <= 1e-3 * e.abs()).all(), ((g - e).abs() / e.abs()).max()
    assert g[-1] < 0.9 * g[0]
