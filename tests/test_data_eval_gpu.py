"""Window builder and evaluation metrics on the MI355X against the CPU oracle, torch's own
DataLoader order, and the reference's Kaggle metric fixture.  Byte-moving work: bit-exact; metrics: 1e-6 (float64)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import climate_amd
    return climate_amd


@pytest.mark.parametrize("shape", [(40, 5, 48, 72, 2, 6), (9, 3, 5, 7, 1, 4), (5, 2, 8, 8, 2, 12), (1, 5, 16, 24, 2, 3)])
def test_window_builder_bit_exact(amd, shape):
    """Every index incl. the left-padded ones (idx < seq_len - 1), a window longer than the data set, odd frame sizes
    (scalar copy path), repeated and unsorted indices."""
    from climate_amd.data import DeviceWindowDataset
    n, c, h, w, co, T = shape
    gen = torch.Generator("cpu").manual_seed(5)
    inp = torch.randn(n, c, h, w, generator=gen)
    out = torch.randn(n, co, h, w, generator=gen)
    ds = DeviceWindowDataset(inp, out, T)
    assert len(ds) == n
    idx = list(range(n)) + [n - 1, 0, n // 2, 0]
    x, y = ds.batch(idx)
    xr, yr = oracle.data_ref.window_batch(inp, out, idx, T)
    assert torch.equal(x.cpu(), xr) and torch.equal(y.cpu(), yr)
    xi, yi = ds[0]
    assert torch.equal(xi.cpu(), xr[0]) and torch.equal(yi.cpu(), yr[0])


def test_device_loader_visits_the_dataloaders_batches(amd):
    """Same generator => same shuffled batches as torch.utils.data.DataLoader over a host data set with the reference's
    __getitem__; full batches are gathered straight into the trainer-style static buffers."""
    from torch.utils.data import DataLoader, Dataset
    from climate_amd.data import DeviceLoader, DeviceWindowDataset
    n, c, h, w, co, T, B = 23, 5, 8, 16, 2, 6, 4
    gen = torch.Generator("cpu").manual_seed(9)
    inp = torch.randn(n, c, h, w, generator=gen)
    out = torch.randn(n, co, h, w, generator=gen)

    class Host(Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            x, y = oracle.data_ref.window_batch(inp, out, [i], T)
            return x[0], y[0]
    ref = list(DataLoader(Host(), batch_size=B, shuffle=True, generator=torch.Generator().manual_seed(1234)))
    got = list((x.cpu().clone(), y.cpu().clone()) for x, y in
               DeviceLoader(DeviceWindowDataset(inp, out, T), B, shuffle=True,
                            generator=torch.Generator().manual_seed(1234)))
    assert len(ref) == len(got) == (n + B - 1) // B
    for (xr, yr), (xg, yg) in zip(ref, got):
        assert torch.equal(xr, xg) and torch.equal(yr, yg)


def test_evaluator_matches_oracle_over_batches(amd):
    from climate_amd.evaluation import DeviceEvaluator, kaggle_score
    stats = {0: {"method": "zscore", "params": {"mean": 281.5, "std": 14.25}},
             1: {"method": "log1p", "params": {"mean": 0.71, "std": 0.52}}}
    h, w, n = 48, 72, 121
    lats = np.linspace(-88.75, 88.75, h)
    gen = torch.Generator("cpu").manual_seed(3)
    pred = torch.randn(n, 2, h, w, generator=gen)
    true = pred + 0.3 * torch.randn(n, 2, h, w, generator=gen)
    ev = DeviceEvaluator(["tas", "pr"], stats, lats, h, w)
    for s in range(0, n, 32):
        ev.update(pred[s:s + 32].cuda(), true[s:s + 32].cuda())
    res = ev.compute("val")
    want = oracle.data_ref.climate_metrics(oracle.data_ref.inverse_transform(pred.double().numpy(), stats),
                                           oracle.data_ref.inverse_transform(true.double().numpy(), stats), lats)
    for i, v in enumerate(("tas", "pr")):
        for k, name in enumerate(("avg/monthly_rmse", "time_mean_rmse", "time_stddev_mae")):
            assert abs(res[f"val/{v}/{name}"] - want[i, k]) < 1e-6 * abs(want[i, k]), (v, name)
    assert abs(res["val/kaggle_score"] - kaggle_score({"tas": want[0], "pr": want[1]})) < 1e-6
    # test split: targets arrive in physical units (main_final.py:454-459)
    ev2 = DeviceEvaluator(["tas", "pr"], stats, lats, h, w, targets_normalized=False)
    phys = torch.from_numpy(oracle.data_ref.inverse_transform(true.double().numpy(), stats)).float()
    ev2.update(pred.cuda(), phys.cuda())
    r2 = ev2.compute("test")
    assert abs(r2["test/tas/time_mean_rmse"] - want[0, 1]) < 1e-4 * want[0, 1]      # (targets rounded to fp32 first)


def test_evaluator_reproduces_the_kaggle_metric_fixture(amd):
    """The reference's one pinned function: _climate_kaggle_metric.score on the fields of _test_kaggle_metric.py."""
    from climate_amd.evaluation import DeviceEvaluator
    g = load_golden("kaggle_metric.npz")
    lats = np.round(np.asarray(g["lats"]), 2)
    pred = torch.stack([torch.as_tensor(g["tas_pred"]), torch.as_tensor(g["pr_pred"])], 1).float()
    true = torch.stack([torch.as_tensor(g["tas_true"]), torch.as_tensor(g["pr_true"])], 1).float()
    ev = DeviceEvaluator(["tas", "pr"], {}, lats, pred.shape[2], pred.shape[3], targets_normalized=False)   # pass-through
    ev.update(pred.cuda(), true.cuda())
    res = ev.compute("test")
    assert abs(res["test/kaggle_score"] - float(g["score"])) < 2e-6 * float(g["score"])       # fp32 inputs vs float64
