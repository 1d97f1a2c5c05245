"""Block-level parity on the MI355X against the reference's own fixtures: ConvLSTM (src/convlstm.py:21-35, BPTT with
gradient on the last step only and on every step) and Up (src/unet.py:60-69), composed by the engine's schedules
from the C-ABI launchers.  Tolerance: relative L2 <= 1e-4."""
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def engine():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climate_amd import engine as e
    return e


class _Serial:
    """engine side-stream interface, serial"""

    @staticmethod
    def run(fn, *tensors):
        fn()


@pytest.mark.parametrize("name", ["convlstm.npz", "convlstm_alldy.npz"])
def test_convlstm_fixture(engine, name):
    """ConvLSTM(16, 8), T=3, B=2 @6x9.  convlstm.npz: only the last hidden state receives gradient (what the model
    does, src/unet_convlstm_attention.py:88); convlstm_alldy.npz: every step does (generic BPTT)."""
    g = load_golden(name)
    xs, dy = g["x_seq"], g["dy"]                                  # [T,B,Cx,h,w], [T,B,Ch,h,w]
    T, B, cx, h, w = xs.shape
    ch = dy.shape[2]
    p = {"convlstm.cell.conv.weight": g["w"].cuda(), "convlstm.cell.conv.bias": g["b"].cuda()}
    gr = {k: torch.zeros_like(v) for k, v in p.items()}
    s4 = xs.permute(1, 0, 2, 3, 4).contiguous().view(B * T, cx, h, w).cuda()     # folded batch: n = b*T + t
    pk = engine.get_plan(p, None, True).pack()
    bott, ctx = engine.convlstm_fwd(p, pk, s4, B, T, save=True)
    h_seq = torch.stack([ctx.hprev[:, t + 1] if t + 1 < T else bott for t in range(T)])
    assert rel_l2(h_seq, g["h_seq"]) < TOL
    plan = engine.get_plan(p, gr, True)
    plan.zero_staging()
    dyd = dy.cuda()
    last_only = bool((dy[:T - 1] == 0).all())
    if last_only:
        ds4 = engine.convlstm_bwd(p, pk, gr, plan.gw, _Serial, ctx, dyd[T - 1].contiguous())
    else:
        ds4 = engine.convlstm_bwd(p, pk, gr, plan.gw, _Serial, ctx, None,
                                  dh_all_steps=dyd.permute(1, 0, 2, 3, 4).contiguous())
    plan.unpack()
    dx_seq = ds4.view(B, T, cx, h, w).permute(1, 0, 2, 3, 4)
    assert rel_l2(dx_seq, g["dx_seq"]) < TOL
    assert rel_l2(gr["convlstm.cell.conv.weight"], g["dw"]) < TOL
    assert rel_l2(gr["convlstm.cell.conv.bias"], g["db"]) < TOL


def test_up_block_fixture(engine):
    """Up(32, 32, 16): ConvTranspose2d(2, s2) -> virtual cat([up, skip]) -> ConvBlock, forward + every gradient."""
    g = load_golden("up_block.npz")
    p = {"u." + k[2:]: v.cuda() for k, v in g.items() if k.startswith("p.")}
    gr = {k: torch.zeros_like(v) for k, v in p.items()}
    x, skip, dy = g["x"].cuda(), g["skip"].cuda(), g["dy"].cuda()
    pk = engine.get_plan(p, None, True).pack()
    out, saved = engine.up_fwd(p, pk, "u.", x, skip, save=True)
    assert rel_l2(out, g["y"]) < TOL
    plan = engine.get_plan(p, gr, True)
    plan.zero_staging()
    dx, dcat = engine.up_bwd(p, pk, gr, plan.gw, _Serial, "u.", saved, dy)
    plan.unpack()
    c_up = p["u.up.weight"].shape[1]
    assert rel_l2(dx, g["dx"]) < TOL
    assert rel_l2(dcat[:, c_up:], g["dskip"]) < TOL
    for k in gr:
        assert rel_l2(gr[k], g["g." + k[2:]]) < TOL, k
