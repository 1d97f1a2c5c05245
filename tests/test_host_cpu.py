"""CPU-side checks: the C-ABI library loads and exports everything include/climate_hip.h declares, the drop-in module
has the reference's exact parameter inventory and default init, the factory / config surface behaves like the
reference's, and the product path refuses to run without the GPU."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, rel_l2


def test_cabi_library_exports_every_declared_symbol():
    from climate_amd._lib import lib, parse_header, LIB_PATH
    assert os.path.exists(LIB_PATH), "build with python physics-based-climate-model_amd/build.py"
    protos = parse_header()
    assert len(protos) >= 37
    dll = lib.load()
    for name in protos:
        assert hasattr(dll, name), name
    assert lib.cm_version() == 3 and lib.cm_arch() == b"gfx950"
    # pure host-side helpers work without a GPU
    assert lib.cm_conv3x3_packed_elems(5, 32) == 16 * 9 * 32
    assert 0 <= lib.cm_conv3x3_pick_config(192, 48, 72, 32) < lib.cm_conv3x3_num_configs()
    assert 0 <= lib.cm_wgrad3x3_pick_config(192, 6, 9, 256) < lib.cm_wgrad3x3_num_configs()


def test_cabi_argument_errors_are_reported_not_crashed():
    from climate_amd._lib import lib
    assert lib.cm_conv3x3(None, 0, 0, None, 0, 0, None, None, None, 0, None, 0, 0, 8, 8, 8, -1, None) == -22
    assert lib.cm_gn_silu_fwd(None, None, None, None, None, None, 2, 12, 64, 8, 1e-5, None) == -22   # 12 % 8 != 0
    assert lib.cm_maxpool2_fwd(None, None, 4, 7, 8, None) == -22                                   # odd height
    assert lib.cm_head_fwd(None, 0, None, None, None, 1, 8, 9, 64, None) == -22                     # out_ch > 8
    # entry points added for the bf16x6 / first-layer / fused kernels validate before touching the device, too
    assert lib.cm_conv3x3_split(None, 0, 24, None, 0, 8, None, None, None, 0, None, 0, 2, 8, 8, 8, 0, None) == -22   # c0 % 16
    assert lib.cm_wgrad3x3_split(None, 0, 24, None, 0, 8, None, 0, None, 32, 0, 2, 8, 8, 8, 0, None) == -22          # c0 % 32
    assert lib.cm_wgrad3x3_smallc(None, 0, 8, None, 0, None, 8, 0, 2, 8, 8, 8, None, None) == -22                    # cin*9 > 64
    assert lib.cm_conv3x3_smallc(None, 0, 8, None, None, None, 0, 2, 8, 8, 8, None) == -22                           # cin*9 > 64
    assert lib.cm_spatial_apply(None, None, None, None, None, None, 1, 2, 4, 7, 8, None) == -22     # pooled output, odd height
    assert lib.cm_conv7_bwd(None, None, None, None, None, None, 2, 8, 8, None) == -22               # no scratch
    assert lib.cm_head_mse_bwd(None, 0, None, None, None, None, None, None, 0, None, None, 1, 8, 9, 64, None) == -22
    assert lib.cm_conv7_bwd_scratch_elems(4, 20) == 4 * 3 * 98
    assert lib.cm_wgrad3x3_smallc_scratch_elems(2, 8, 8, 40) == 2 * 2 * 32 * 64


@pytest.mark.parametrize("base,in_ch", [(8, 5), (16, 7), (32, 5)])
def test_state_dict_matches_reference_inventory(base, in_ch):
    from climate_amd.model import AttUNetConvLSTM
    m = AttUNetConvLSTM(in_ch=in_ch, out_ch=2, base=base, seq_len=6)
    want = oracle.param_shapes(in_ch, 2, base)
    sd = m.state_dict()
    assert list(sd) == list(want) and len(sd) == 75
    for k, s in want.items():
        assert tuple(sd[k].shape) == s
    assert [n for n, _ in m.named_parameters()] == list(want)
    assert len(m._grad_names) == 73 and not any(n.startswith("post_conv") for n in m._grad_names)


def test_default_init_equals_reference_under_seed():
    """torch.manual_seed(42) + construction consumes the RNG exactly like the reference's __init__ does."""
    from climate_amd.model import AttUNetConvLSTM
    g = load_golden("init_seed42_base32.npz")
    torch.manual_seed(42)
    m = AttUNetConvLSTM(in_ch=5, out_ch=2, base=32, seq_len=6)
    sd = m.state_dict()
    assert list(sd) == g["names"].tolist()
    for k, s, f in zip(g["names"].tolist(), g["sums"].tolist(), g["firsts"]):
        assert abs(sd[k].double().sum().item() - s) < 1e-9, k
        n = min(4, sd[k].numel())
        assert np.allclose(sd[k].flatten()[:n].double().numpy(), np.asarray(f)[:n], atol=0)


def test_reference_checkpoint_layout_loads_both_ways():
    from climate_amd.model import AttUNetConvLSTM
    p = oracle.closed_form_params(5, 2, 8)
    m = AttUNetConvLSTM(5, 2, 8)
    missing = m.load_state_dict(p, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    flat = m.flatten_parameters_()
    assert flat.numel() >= sum(v.numel() for v in p.values())
    for k, v in m.state_dict().items():                 # values survive flattening, views alias the flat buffer
        assert torch.equal(v, p[k])
    m.enc1.body[0].weight.data.add_(1.0)
    assert torch.equal(flat[:m.enc1.body[0].weight.numel()].view_as(p["enc1.body.0.weight"]),
                       p["enc1.body.0.weight"] + 1.0)
    # Lightning prefixes keys with "model."
    from climate_amd.lightning_module import ClimateEmulationModule
    lm = ClimateEmulationModule(AttUNetConvLSTM(5, 2, 8), learning_rate=5e-4, weight_decay=0.0)
    assert all(k.startswith("model.") for k in lm.state_dict())
    assert len(lm.state_dict()) == 75


def test_factory_and_config_surface():
    import climate_amd
    from climate_amd.config import load_config, synthetic_config
    from climate_amd.model import AttUNetConvLSTM, get_model
    cfg = load_config(os.path.join(climate_amd._PKG_DIR, "configs"),
                      overrides=["model.base_channels=16", "training.lr=1e-3", "trainer.devices=8"])
    assert cfg.model.type == "unet_convlstm_attention" and cfg.model.base_channels == 16
    assert cfg.training.lr == 1e-3 and cfg.training.weight_decay == 0 and cfg.trainer.devices == 8 and cfg.seed == 42
    m = get_model(cfg)
    assert isinstance(m, AttUNetConvLSTM) and m.in_ch == 5 and m.out_ch == 2 and m.base == 16 and m.seq_len == 6
    cfg.model["in_ch"] = 7                               # the reference's hard-coded value, opt-in
    assert get_model(cfg).in_ch == 7
    cfg.model["type"] = "does_not_exist"
    with pytest.raises(ValueError, match="Unknown model type: does_not_exist"):
        get_model(cfg)
    assert get_model(synthetic_config(base_channels=32)).base == 32


def test_factory_serves_every_reference_model_type():
    """src/models.py:7-38 dispatches on four model.type values; replacing the reference's factory must not break any
    of them (SimpleCNN = BASELINE configs[0], on the HIP path since round 3; unet = the same HIP kernels on one frame)."""
    import climate_amd
    from climate_amd.config import load_config
    from climate_amd.simple_cnn import SimpleCNN
    from climate_amd.model import UNet, get_model
    cdir = os.path.join(climate_amd._PKG_DIR, "configs")
    g = load_golden("simple_cnn_default_cfg.npz")
    m = get_model(load_config(cdir, overrides=["model=SimpleCNN"]))
    assert isinstance(m, SimpleCNN)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 10730626
    assert list(m.state_dict()) == g["names"].tolist()
    with pytest.raises(RuntimeError, match="HIP path only"):                           # no CPU fallback
        m.eval()(torch.zeros(2, 5, 16, 24))
    u = get_model(load_config(cdir, overrides=["model=unet", "model.base_channels=8"]))
    assert isinstance(u, UNet) and u.base == 8
    want = oracle.unet_param_shapes(5, 2, 8)
    assert list(u.state_dict()) == list(want) == load_golden("unet_tiny.npz")["names"].tolist()
    assert all(tuple(v.shape) == want[k] for k, v in u.state_dict().items())
    assert u._grad_names == list(want)                                                   # every parameter is trained
    with pytest.raises(RuntimeError, match="HIP path only"):
        u(torch.zeros(1, 5, 8, 8))


def test_cnn_transformer_matches_reference():
    """model.type = cnn_transformer (src/cnn_transformer.py:4-54; BASELINE configs[3]): the HIP module ``get_model``
    serves has the reference's state_dict and default init under a seed (as far as a CPU can see: names, shapes, initial
    values, constructor checks), and the CPU oracle reproduces the reference's eval forward and gradients."""
    import climate_amd
    from climate_amd.config import load_config
    from climate_amd.model import get_model
    g = load_golden("cnn_transformer_tiny.npz")
    torch.manual_seed(42)
    from climate_amd.cnn_transformer import CNNTransformer as HipCNNTransformer
    m = HipCNNTransformer(in_channels=5, out_channels=2, embed_dim=32, depth=2, n_heads=4, mlp_dim=48, dropout=0.1)
    sd = m.state_dict()
    assert list(sd) == g["names"].tolist()
    for k, sm in zip(g["names"].tolist(), g["sums"].tolist()):
        assert abs(sd[k].double().sum().item() - sm) < 1e-9, k
    P = {k: v.detach().clone().requires_grad_() for k, v in sd.items()}
    x = g["x"].clone().requires_grad_()
    y = oracle.cnn_transformer_forward(P, x, 4); y.square().mean().backward()
    assert torch.allclose(y, g["y_eval"], rtol=0, atol=2e-6)
    assert torch.allclose(x.grad, g["dx"], rtol=1e-4, atol=1e-8)
    assert torch.allclose(P["pos_embedding"].grad, g["g_pos"], rtol=1e-4, atol=1e-8)
    assert torch.allclose(P["transformer.layers.0.self_attn.in_proj_weight"].grad, g["g_inproj0"], rtol=1e-4, atol=1e-8)
    big = get_model(load_config(os.path.join(climate_amd._PKG_DIR, "configs"),
                                overrides=["model=cnn_transformer", "model.embed_dim=256", "model.depth=6",
                                           "model.n_heads=8"]))
    c4 = load_golden("cnn_transformer_cfg4.npz")
    assert isinstance(big, HipCNNTransformer) and list(big.state_dict()) == c4["names"].tolist()
    assert sum(p.numel() for p in big.parameters()) == int(c4["n_params"]) == 2895170
    with pytest.raises(RuntimeError, match="HIP path only"):
        big(torch.zeros(1, 5, 48, 72))
    # same constructor, same draws from the global generator, same state_dict as the reference's module
    torch.manual_seed(42)
    h = HipCNNTransformer(in_channels=5, out_channels=2, embed_dim=32, depth=2, n_heads=4, mlp_dim=48, dropout=0.1)
    assert list(h.state_dict()) == g["names"].tolist()
    for k in sd:
        assert torch.equal(h.state_dict()[k], sd[k]), k
    with pytest.raises(ValueError, match="head_dim"):
        HipCNNTransformer(embed_dim=128, n_heads=2)


def test_simple_cnn_matches_reference():
    """SimpleCNN (src/models.py:44-123) against the fixture generated from the reference: the HIP module has the reference's
    state_dict and default init under a seed; the CPU oracle reproduces the eval forward, the train-mode forward / backward
    (BatchNorm batch statistics, the Dropout2d multipliers the reference drew) and the running-buffer updates."""
    from climate_amd.simple_cnn import SimpleCNN
    g = load_golden("simple_cnn.npz")
    torch.manual_seed(42)
    m = SimpleCNN(n_input_channels=5, n_output_channels=2, kernel_size=3, init_dim=8, depth=3, dropout_rate=0.2)
    sd = m.state_dict()
    assert list(sd) == g["names"].tolist()
    for k, shp, sm in zip(g["names"].tolist(), g["shapes"], g["sums"].tolist()):
        assert tuple(sd[k].shape) == tuple(int(v) for v in shp[:sd[k].dim()])
        assert abs(sd[k].double().sum().item() - sm) < 1e-9, k
    assert m._grad_names == [n for n, _ in m.named_parameters()]                     # every parameter is trained
    P = {k: v.detach().clone().requires_grad_() for k, v in m.named_parameters()}
    B = {k: v.detach().clone() for k, v in m.named_buffers()}
    x = g["x"]
    assert torch.allclose(oracle.simple_cnn_forward(P, B, x, training=False), g["y_eval"], rtol=0, atol=1e-6)
    xg = x.clone().requires_grad_()
    y = oracle.simple_cnn_forward(P, B, xg, training=True, drop_mask=g["drop_mask"]); y.square().mean().backward()
    assert torch.allclose(y, g["y_train"], rtol=0, atol=1e-6)
    assert torch.allclose(xg.grad, g["dx_train"], rtol=1e-5, atol=1e-7)
    for k in P:
        assert rel_l2(P[k].grad, g["g." + k]) < 1e-5 or g["g." + k].abs().max() < 1e-6, k
    for k in B:
        if "running_" in k:
            assert torch.allclose(B[k], g["b." + k], rtol=1e-6, atol=1e-7), k
    with pytest.raises(ValueError, match="kernel_size"):
        SimpleCNN(5, 2, kernel_size=5)


def test_hip_adam_state_interoperates_with_torch_adam():
    """HipAdam's optimizer state has torch.optim.Adam's layout: a reference checkpoint's optimizer state
    loads into HipAdam and vice versa (Lightning restores it on ckpt_path= resume, main_final.py:770-774)."""
    from climate_amd.model import AttUNetConvLSTM
    from climate_amd.optim import HipAdam
    torch.manual_seed(0)
    m = AttUNetConvLSTM(5, 2, 8)
    ref = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=0.0)
    for n, p in m.named_parameters():
        if not n.startswith("post_conv."):
            p.grad = torch.randn_like(p)
    ref.step(); ref.step()
    sd = ref.state_dict()
    ours = HipAdam(m.parameters(), lr=1e-3)
    ours.load_state_dict(sd)                                   # reference -> ours
    assert ours.param_groups[0]["lr"] == 5e-4
    named = dict(m.named_parameters())
    st = ours.state[named["enc1.body.0.weight"]]
    assert torch.is_tensor(st["step"]) and float(st["step"]) == 2.0
    assert st["exp_avg"].shape == named["enc1.body.0.weight"].shape
    assert named["post_conv.0.weight"] not in ours.state or not ours.state[named["post_conv.0.weight"]]
    back = torch.optim.Adam(m.parameters(), lr=1e-3)
    back.load_state_dict(ours.state_dict())                    # ours -> reference
    for a, b in zip(sd["state"].values(), back.state_dict()["state"].values()):
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])
        assert float(a["step"]) == float(b["step"])
    assert set(sd["param_groups"][0]) == set(ours.state_dict()["param_groups"][0])


def test_flat_adam_state_roundtrip_against_torch_adam():
    """The fused trainer's flat moment buffers <-> torch.optim.Adam's state_dict (pure host logic, CPU tensors)."""
    from climate_amd.model import AttUNetConvLSTM
    from climate_amd.optim import flat_adam_state_dict, load_flat_adam_state
    torch.manual_seed(1)
    m = AttUNetConvLSTM(5, 2, 8)
    names = [n for n, _ in m.named_parameters()]
    ref = torch.optim.Adam(m.parameters(), lr=5e-4)
    for n, p in m.named_parameters():
        if not n.startswith("post_conv."):
            p.grad = torch.randn_like(p)
    for _ in range(3):
        ref.step()
    sd = ref.state_dict()
    lay, nt = m._build_layout(), m.n_flat_trainable
    fm, fv = torch.full((nt,), 9.0), torch.full((nt,), 9.0)
    assert load_flat_adam_state(sd, lay, names, fm, fv) == 3
    o, k, shp = lay["up2.conv.body.3.weight"]
    idx = names.index("up2.conv.body.3.weight")
    assert torch.equal(fm[o:o + k].view(shp), sd["state"][idx]["exp_avg"])
    out = flat_adam_state_dict(lay, names, m._grad_names, fm, fv, 3, 5e-4, (0.9, 0.999), 1e-8, 0.0)
    assert set(out["state"]) == set(sd["state"]) and len(out["state"]) == 73
    for i in sd["state"]:
        assert torch.equal(out["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
        assert float(out["state"][i]["step"]) == 3.0
    fresh = torch.optim.Adam(m.parameters(), lr=1e-3)
    fresh.load_state_dict(out)                                 # torch accepts what the fused trainer writes
    assert fresh.param_groups[0]["lr"] == 5e-4
    assert flat_adam_state_dict(lay, names, m._grad_names, fm, fv, 0, 5e-4, (0.9, 0.999), 1e-8, 0.0)["state"] == {}


def test_plan_cache_keeps_referenced_plans():
    """engine.get_plan's LRU never evicts a plan something else still holds (captured graphs point into its arenas)."""
    import sys
    from climate_amd import engine
    assert engine._PLANS_MAX >= 2
    src = open(engine.__file__).read()
    assert "getrefcount" in src and "_PLANS.clear()" not in src


def test_product_path_has_no_cpu_fallback():
    from climate_amd.model import AttUNetConvLSTM
    from climate_amd import ops
    m = AttUNetConvLSTM(5, 2, 8)
    with pytest.raises(RuntimeError, match="HIP path only"):
        m(torch.zeros(1, 2, 5, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.maxpool2_fwd(torch.zeros(1, 1, 4, 4))


def test_lightning_module_surface():
    from climate_amd.lightning_module import ClimateEmulationModule
    from climate_amd.model import AttUNetConvLSTM
    from climate_amd.optim import HipAdam
    lm = ClimateEmulationModule(AttUNetConvLSTM(5, 2, 8), learning_rate=5e-4, weight_decay=0.0)
    opt = lm.configure_optimizers()
    assert isinstance(opt, HipAdam) and isinstance(opt, torch.optim.Optimizer)
    g = opt.param_groups[0]
    assert g["lr"] == 5e-4 and g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8 and g["weight_decay"] == 0.0
    assert len(g["params"]) == 75
    for name in ("forward", "training_step", "configure_optimizers"):
        assert callable(getattr(lm, name))


def test_gradient_buckets_partition_the_flat_buffer():
    """Two-bucket exchange (trainer.py): the encoder's gradients are exactly the prefix [0, bucket_boundary) of the
    flat buffer, ConvLSTM + decoder + head exactly the rest of the trainable range; the staged weight-gradient unpack
    jobs split the same way."""
    from climate_amd.model import AttUNetConvLSTM
    for base in (8, 32, 64):
        m = AttUNetConvLSTM(5, 2, base, 6)
        lay, b, nt = m._build_layout(), m.bucket_boundary, m.n_flat_trainable
        assert 0 < b < nt and b % 64 == 0
        for name in m._grad_names:
            o, k, _ = lay[name]
            assert (o + k <= b) == name.startswith("enc"), name
            assert o + k <= nt
        assert all(lay[n][0] >= nt for n in lay if n.startswith("post_conv."))
