"""Pins the CPU oracle (oracle/cpu_ref.py) against fixtures produced by the real reference modules
(tests/golden/gen_golden.py).  Pure CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import load_golden, rel_l2

TOL = 2e-6      # oracle and reference are the same fp32 maths on the same CPU kernels


def _sub(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_param_inventory_matches_reference_names():
    g = load_golden("model_tiny.npz")
    in_ch, out_ch, base = (int(v) for v in g["cfg"][:3])
    shapes = oracle.param_shapes(in_ch, out_ch, base)
    ref = _sub(g, "p1.")
    assert list(shapes) == list(ref)
    assert len(shapes) == 75
    for k, s in shapes.items():
        assert tuple(ref[k].shape) == s
    assert sorted(g["nograd"].tolist()) == ["post_conv.0.bias", "post_conv.0.weight"]


def test_se_block():
    g = load_golden("se_block.npz")
    x = g["x"].clone().requires_grad_(); w1 = g["w1"].clone().requires_grad_(); w2 = g["w2"].clone().requires_grad_()
    y = oracle.se_block(x, w1, w2); y.backward(g["dy"])
    for got, want in ((y, g["y"]), (x.grad, g["dx"]), (w1.grad, g["dw1"]), (w2.grad, g["dw2"])):
        assert rel_l2(got, want) < TOL


def test_spatial_gate_with_ties():
    g = load_golden("spatial_gate.npz")
    x = g["x"].clone().requires_grad_(); w = g["w7"].clone().requires_grad_()
    y = oracle.spatial_gate(x, w); y.backward(g["dy"])
    for got, want in ((y, g["y"]), (x.grad, g["dx"]), (w.grad, g["dw7"])):
        assert rel_l2(got, want) < TOL


def test_maxpool_tie_semantics():
    g = load_golden("maxpool.npz")
    x = g["x"].clone().requires_grad_()
    y = F.max_pool2d(x, 2); y.backward(g["dy"])
    assert torch.equal(y, g["y"]) and torch.equal(x.grad, g["dx"])
    # all-equal window -> first element in scan order gets the whole gradient
    assert x.grad[0, 0, 0, 0] == g["dy"][0, 0, 0, 0] and x.grad[0, 0, 0, 1] == 0 and x.grad[0, 0, 1, 0] == 0


def test_conv_block():
    g = load_golden("conv_block.npz")
    p = {k: v.clone().requires_grad_() for k, v in _sub(g, "p.").items()}
    x = g["x"].clone().requires_grad_()
    y = oracle.conv_block(x, p, ""); y.backward(g["dy"])
    assert rel_l2(y, g["y"]) < TOL and rel_l2(x.grad, g["dx"]) < 1e-5
    for k, want in _sub(g, "g.").items():
        assert rel_l2(p[k].grad, want) < 1e-5, k


def test_up_block():
    g = load_golden("up_block.npz")
    p = {k: v.clone().requires_grad_() for k, v in _sub(g, "p.").items()}
    x = g["x"].clone().requires_grad_(); sk = g["skip"].clone().requires_grad_()
    y = oracle.up_block(x, sk, p, ""); y.backward(g["dy"])
    assert rel_l2(y, g["y"]) < TOL and rel_l2(x.grad, g["dx"]) < 1e-5 and rel_l2(sk.grad, g["dskip"]) < 1e-5
    for k, want in _sub(g, "g.").items():
        assert rel_l2(p[k].grad, want) < 1e-5, k


@pytest.mark.parametrize("name", ["convlstm.npz", "convlstm_alldy.npz"])
def test_convlstm(name):
    g = load_golden(name)
    xs = g["x_seq"].clone().requires_grad_(); w = g["w"].clone().requires_grad_(); b = g["b"].clone().requires_grad_()
    hs = oracle.convlstm(xs, w, b); hs.backward(g["dy"])
    assert rel_l2(hs, g["h_seq"]) < TOL
    for got, want in ((xs.grad, g["dx_seq"]), (w.grad, g["dw"]), (b.grad, g["db"])):
        assert rel_l2(got, want) < 1e-5


@pytest.mark.parametrize("name", ["model_tiny.npz", "model_tiny_b16.npz"])
def test_whole_model_forward_backward(name):
    g = load_golden(name)
    in_ch, out_ch, base = (int(v) for v in g["cfg"][:3])
    salt = int(g["salt"]) if "salt" in g else 0
    p = {k: v.clone().requires_grad_() for k, v in oracle.closed_form_params(in_ch, out_ch, base, salt=salt).items()}
    x = g["x"].clone().requires_grad_()
    loss = oracle.training_loss(p, x, g["y"]); loss.backward()
    assert abs(loss.item() - float(g["loss1"])) <= 1e-6 * abs(float(g["loss1"]))
    assert rel_l2(oracle.model_forward(p, x).detach(), g["pred"]) < TOL
    assert rel_l2(x.grad, g["dx"]) < 1e-5
    for k, want in _sub(g, "g.").items():
        assert rel_l2(p[k].grad, want) < 1e-5, k
    assert p["post_conv.0.weight"].grad is None and p["post_conv.0.bias"].grad is None


def test_three_adam_steps():
    g = load_golden("model_tiny.npz")
    in_ch, out_ch, base = (int(v) for v in g["cfg"][:3])
    p = {k: v.clone().requires_grad_() for k, v in oracle.closed_form_params(in_ch, out_ch, base).items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(1, 4):
        for t in p.values():
            t.grad = None
        loss = oracle.training_loss(p, g["x"], g["y"]); loss.backward()
        assert abs(loss.item() - float(g[f"loss{step}"])) <= 2e-6 * abs(float(g[f"loss{step}"]))
        with torch.no_grad():
            for k, t in p.items():
                if t.grad is not None:
                    oracle.adam_reference_step(t, t.grad, m[k], v2[k], step, lr=5e-4)
        if step in (1, 3):
            for k, want in _sub(g, f"p{step}.").items():
                assert rel_l2(p[k].detach(), want) < 1e-6, (step, k)


@pytest.mark.slow
def test_cfg2_checksums():
    """BASELINE config 2 shape on CPU (a few seconds): loss, output norm and per-parameter gradient norms."""
    g = load_golden("cfg2_checksums.npz")
    in_ch, out_ch, base, T, B, H, W = (int(v) for v in g["cfg"])
    p = {k: v.clone().requires_grad_() for k, v in oracle.closed_form_params(in_ch, out_ch, base).items()}
    gen = torch.Generator("cpu").manual_seed(int(g["seed"]))
    x = torch.randn(B, T, in_ch, H, W, generator=gen); y = torch.randn(B, out_ch, H, W, generator=gen)
    pred = oracle.model_forward(p, x); loss = F.mse_loss(pred, y); loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert abs(pred.double().norm().item() - float(g["pred_l2"])) < 1e-5 * float(g["pred_l2"])
    for name, want in zip(g["grad_names"].tolist(), g["grad_l2"].tolist()):
        assert abs(p[name].grad.double().norm().item() - want) <= 5e-5 * want + 1e-12, name


def test_default_init_left_padded_window_vs_reference():
    """Default init (beta = 0) + all-zero leading frames (main_final.py:76,127-131): whole frames of exact C-way amax
    ties and all-equal MaxPool windows.  Oracle forward, loss and all 73 gradients vs the reference's."""
    g = load_golden("model_default_init_padded.npz")
    p = {k: v.clone().requires_grad_() for k, v in _sub(g, "p.").items()}
    x = g["x"].clone().requires_grad_()
    pred = oracle.model_forward(p, x)
    assert rel_l2(pred, g["pred"]) < TOL
    loss = F.mse_loss(pred, g["y"]); loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * float(g["loss"])
    assert rel_l2(x.grad, g["dx"]) < 1e-5
    for k, want in _sub(g, "g.").items():
        assert rel_l2(p[k].grad, want) < 1e-5, k
    assert p["post_conv.0.weight"].grad is None


def test_plain_unet_vs_reference():
    g = load_golden("unet_tiny.npz")
    in_ch, out_ch, base = (int(v) for v in g["cfg"][:3])
    P = oracle.closed_form_params(in_ch, out_ch, base, salt=int(g["salt"]),
                                  shapes=oracle.unet_param_shapes(in_ch, out_ch, base))
    assert list(P) == g["names"].tolist()
    p = {k: v.clone().requires_grad_() for k, v in P.items()}
    x = g["x"].clone().requires_grad_()
    pred = oracle.unet_forward(p, x)
    assert rel_l2(pred, g["pred"]) < TOL
    F.mse_loss(pred, g["y"]).backward()
    assert rel_l2(x.grad, g["dx"]) < 1e-5
    for k, want in _sub(g, "g.").items():
        assert rel_l2(p[k].grad, want) < 1e-5, k


def test_imposed_decisions_are_neutral_when_they_are_the_oracles_own():
    """oracle.Decisions with the oracle's own amax masks / MaxPool choices reproduces the plain gradient exactly and
    reports no differing / violating site; a wrong choice is counted as a violation."""
    import oracle.cpu_ref as ref
    P = {k: v.double().requires_grad_() for k, v in oracle.closed_form_params(5, 2, 8).items()}
    gen = torch.Generator("cpu").manual_seed(3)
    x = torch.randn(2, 3, 5, 16, 24, generator=gen).double()
    y = torch.randn(2, 2, 16, 24, generator=gen).double()
    oracle.training_loss(P, x, y).backward()
    g0 = {k: v.grad.clone() for k, v in P.items() if v.grad is not None}
    gate_in, pool_in = [], []
    o_sg, o_mp = ref.spatial_gate, ref._max_pool

    def sg(xx, w7, dec=None, site=None):
        gate_in.append(xx.detach())
        return o_sg(xx, w7, dec, site)

    def mp(xx, dec, site):
        pool_in.append((site, xx.detach()))
        return o_mp(xx, dec, site)
    ref.spatial_gate, ref._max_pool = sg, mp
    try:
        with torch.no_grad():
            oracle.model_forward(P, x)
    finally:
        ref.spatial_gate, ref._max_pool = o_sg, o_mp
    sites = [(pfx, t) for t in range(3) for pfx in ("enc1.", "enc2.conv.", "enc3.conv.", "enc4.conv.")]
    sites += [("up3.conv.", None), ("up2.conv.", None), ("up1.conv.", None)]
    dec = oracle.Decisions()
    for site, u in zip(sites, gate_in):
        dec.amax[site] = u == u.amax(1, keepdim=True)
    for site, v in pool_in:
        n, c, h, w = v.shape
        win = v.view(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
        dec.pool[site] = win.argmax(-1)
    for v in P.values():
        v.grad = None
    oracle.training_loss(P, x, y, decisions=dec).backward()
    assert dec.violations == 0 and dec.differing == 0 and dec.sites > 0
    for k in g0:
        assert torch.equal(P[k].grad, g0[k]), k
    bad = oracle.Decisions()
    wrong = dec.amax[("up1.conv.", None)].clone()
    u = gate_in[-1]
    wrong[0, :, 0, 0] = u[0, :, 0, 0] == u[0, :, 0, 0].amin()       # impose the MINIMUM channel at one pixel
    bad.amax[("up1.conv.", None)] = wrong
    oracle.model_forward(P, x, decisions=bad)
    assert bad.violations >= 1
