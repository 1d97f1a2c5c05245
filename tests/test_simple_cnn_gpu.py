"""SimpleCNN (reference src/models.py:44-123, BASELINE.json configs[0]) on the HIP path: BatchNorm2d / Dropout2d / residual
kernels and the whole model against the fixture generated from the reference and against the float64 oracle.

Tolerance: relative L2 <= 1e-4 on outputs and gradients (north_star's fp32 bound); observed ~1e-6.
"""
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climate_amd._lib import lib as l
    return l


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("shape", [(8, 64, 48, 72), (2, 16, 8, 12), (3, 8, 5, 7), (4, 32, 6, 9), (1, 8, 4, 4)])
@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, False)])
def test_batchnorm_kernels(lib, shape, relu, with_res):
    """cm_bn_fwd / cm_bn_bwd vs torch (float64): train-mode statistics + running-buffer update, eval mode, the fused
    residual add and ReLU, the residual gradient."""
    from climate_amd._lib import check
    n, c, h, w = shape
    torch.manual_seed(2)
    x = (torch.randn(n, c, h, w) * 2.0 + 0.7).cuda()
    res = torch.randn(n, c, h, w).cuda() if with_res else None
    gamma = (torch.randn(c) * 0.3 + 1.0).cuda()
    beta = (torch.randn(c) * 0.2).cuda()
    dy = torch.randn(n, c, h, w).cuda()
    for training in (True, False):
        rm = (torch.randn(c) * 0.1).cuda()
        rv = (torch.rand(c) + 0.5).cuda()
        rm0, rv0 = rm.clone(), rv.clone()
        y = torch.empty_like(x)
        save = torch.empty(c, 2, device="cuda")
        check(lib.cm_bn_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None if res is None else res.data_ptr(),
                            y.data_ptr(), save.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, int(relu),
                            int(training), n, c, h * w, _st()), "bn_fwd")
        xd = x.double().cpu().requires_grad_()
        gd = gamma.double().cpu().requires_grad_()
        bd = beta.double().cpu().requires_grad_()
        rd = None if res is None else res.double().cpu().requires_grad_()
        rmd, rvd = rm0.double().cpu(), rv0.double().cpu()
        ref = F.batch_norm(xd, rmd, rvd, gd, bd, training, 0.1, 1e-5)
        if rd is not None:
            ref = ref + rd
        if relu:
            ref = F.relu(ref)
        assert rel_l2(y, ref) < 2e-6
        assert rel_l2(rm, rmd) < 2e-6 and rel_l2(rv, rvd) < 2e-6            # (unchanged in eval mode)
        ref.backward(dy.double().cpu())
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if with_res else None
        dg = torch.zeros(c, device="cuda")
        db = torch.zeros(c, device="cuda")
        check(lib.cm_bn_bwd(x.data_ptr(), y.data_ptr() if relu else None, dy.data_ptr(), gamma.data_ptr(),
                            save.data_ptr(), dx.data_ptr(), None if dres is None else dres.data_ptr(), dg.data_ptr(),
                            db.data_ptr(), int(relu), int(training), n, c, h * w, _st()), "bn_bwd")
        assert rel_l2(dx, xd.grad) < 5e-6
        assert rel_l2(dg, gd.grad) < 5e-6 and rel_l2(db, bd.grad) < 5e-6
        if with_res:
            assert rel_l2(dres, rd.grad) < 1e-6


def test_plane_scaling_and_center_tap(lib):
    from climate_amd._lib import check
    torch.manual_seed(0)
    x = torch.randn(3, 7, 5, 6, device="cuda")
    m = (torch.rand(3, 7, device="cuda") > 0.3).float() * 1.25
    out = torch.empty_like(x)
    check(lib.cm_scale_planes(x.data_ptr(), m.data_ptr(), out.data_ptr(), 21, 30, _st()), "scale")
    assert torch.equal(out, x * m[:, :, None, None])
    w1 = torch.randn(6, 5, 1, 1, device="cuda")
    w3 = torch.full((6, 5, 3, 3), 7.0, device="cuda")
    check(lib.cm_embed_center_tap(w1.data_ptr(), w3.data_ptr(), 6, 5, _st()), "embed")
    ref = torch.zeros(6, 5, 3, 3, device="cuda")
    ref[:, :, 1, 1] = w1[:, :, 0, 0]
    assert torch.equal(w3, ref)
    g = torch.randn(6, 9, 5, device="cuda")
    dw = torch.ones(6, 5, device="cuda")
    check(lib.cm_extract_center_tap(g.data_ptr(), dw.data_ptr(), 6, 5, _st()), "extract")
    assert torch.equal(dw, 1.0 + g[:, 4, :])


def _module(seed=42, **kw):
    from climate_amd.simple_cnn import SimpleCNN
    torch.manual_seed(seed)
    return SimpleCNN(n_input_channels=5, n_output_channels=2, kernel_size=3, **kw)


def test_simple_cnn_reference_fixture(lib):
    """tests/golden/simple_cnn.npz (generated from the reference's SimpleCNN): eval forward, train-mode forward with the
    Dropout2d multipliers the reference drew, d(input), every parameter gradient, the BatchNorm running buffers."""
    g = load_golden("simple_cnn.npz")
    m = _module(init_dim=8, depth=3, dropout_rate=0.2).cuda()
    x = g["x"].cuda()
    m.eval()
    with torch.no_grad():
        assert rel_l2(m(x), g["y_eval"]) < TOL
    m.train()
    m.impose_dropout_mask(g["drop_mask"])
    xg = x.clone().requires_grad_()
    y = m(xg)
    y.square().mean().backward()
    assert rel_l2(y, g["y_train"]) < TOL
    assert rel_l2(xg.grad, g["dx_train"]) < TOL
    assert rel_l2(m.initial[0].weight.grad, g["g_initial0"]) < TOL
    big = max(g["g." + k].norm().item() for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        ref = g["g." + k]
        if ref.norm().item() < 1e-5 * big:
            # a convolution bias in front of a BatchNorm: its gradient is zero in exact arithmetic (the mean is removed)
            # and rounding noise in any implementation, the reference's included
            assert p.grad.norm().item() < 1e-4 * big, k
            continue
        assert rel_l2(p.grad, ref) < TOL, k
    for k, b in m.named_buffers():
        if "running_" in k:
            assert rel_l2(b, g["b." + k]) < 1e-5, k
        elif "num_batches_tracked" in k:
            assert int(b.item()) == 1, k


def _relu_decisions(sv):
    dec = oracle.Decisions(delta=1e-5)
    dec.relu["initial"] = (sv.a0 > 0).cpu()
    for i, blk in enumerate(sv.blocks):
        dec.relu[("res", i, 1)] = (blk[2] > 0).cpu()
        dec.relu[("res", i, 2)] = (blk[8] > 0).cpu()
    dec.relu["final"] = (sv.af > 0).cpu()
    return dec


@pytest.mark.parametrize("cfg", [dict(init_dim=16, depth=3, B=4, H=16, W=24), dict(init_dim=64, depth=4, B=8, H=48, W=72)])
def test_simple_cnn_vs_float64_oracle(lib, cfg):
    """Training-mode loss, d(input) and every gradient against the float64 oracle with the DEVICE's Dropout2d multipliers
    and ReLU decisions imposed (and validated); the second case is BASELINE configs[0] (init_dim 64, depth 4, batch 8,
    48x72).  Also: the device's multipliers are 0 or 1/(1-p) with a plausible keep rate and differ from step to step."""
    from climate_amd.model import _HotPathFunction  # noqa: F401
    B, H, W = cfg["B"], cfg["H"], cfg["W"]
    m = _module(init_dim=cfg["init_dim"], depth=cfg["depth"], dropout_rate=0.2).cuda().train()
    m.reseed_dropout(123)
    torch.manual_seed(5)
    x = torch.randn(B, 5, H, W)
    yt = torch.randn(B, 2, H, W)
    p = m._param_dict()
    P64 = {k: v.detach().double().cpu().requires_grad_() for k, v in p.items()}
    B64 = {k: v.detach().double().cpu() for k, v in m.named_buffers() if "running_" in k}
    from climate_amd import engine
    pk = engine.get_plan(p, None, True).pack()
    pred, sv = m._engine_forward(p, pk, x.cuda(), save=True)
    mult = sv.mult
    vals = set(mult.unique().tolist())
    assert vals <= {0.0, 1.25} and 0.5 < (mult > 0).float().mean().item() < 0.97
    flat = torch.zeros(m.n_flat_trainable, device="cuda")
    gv = m._views(flat)
    dpred = (2.0 * (pred - yt.cuda()) / pred.numel()).contiguous()
    dx = m._engine_backward(p, pk, gv, sv, dpred, need_dx=True)
    xd = x.double().requires_grad_()
    ref = oracle.simple_cnn_forward(P64, B64, xd, training=True, drop_mask=mult.double().cpu(),
                                    decisions=_relu_decisions(sv))
    loss = F.mse_loss(ref, yt.double())
    loss.backward()
    assert rel_l2(pred, ref) < TOL
    assert rel_l2(dx, xd.grad) < TOL
    big = max(v.grad.norm().item() for v in P64.values())
    for k in p:
        r = P64[k].grad
        if r.norm().item() < 1e-5 * big:
            assert gv[k].norm().item() < 1e-4 * big, k
            continue
        assert rel_l2(gv[k], r) < TOL, k
    for k, v in m.named_buffers():
        if "running_" in k:
            assert rel_l2(v, B64[k]) < 1e-5, k
    # a second forward draws different multipliers
    _, sv2 = m._engine_forward(p, pk, x.cuda(), save=True)
    assert not torch.equal(sv2.mult, mult)


def test_simple_cnn_fused_trainer(lib):
    """HotPathTrainer on SimpleCNN: the graph-replayed step equals the eager one (same dropout stream), the loss falls,
    BatchNorm forbids the two-micro-batch schedule."""
    from climate_amd.trainer import HotPathTrainer
    torch.manual_seed(9)
    x = torch.randn(8, 5, 16, 24).cuda()
    y = torch.randn(8, 2, 16, 24).cuda()
    losses = {}
    for graph in (False, True):
        m = _module(init_dim=16, depth=3, dropout_rate=0.2).cuda().train()
        m.reseed_dropout(77)
        tr = HotPathTrainer(m, lr=1e-3, use_graph=graph, distributed=False)
        losses[graph] = [tr.step(x, y).item() for _ in range(4)]
        assert tr._parts == 1
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) <= 2e-5 * abs(a), (losses[False], losses[True])
    assert losses[True][-1] < losses[True][0]
    m = _module(init_dim=16, depth=3, dropout_rate=0.2).cuda().train()
    with pytest.raises(ValueError, match="BatchNorm"):
        HotPathTrainer(m, use_graph=False, distributed=False, micro_batches=2).step(x, y)
