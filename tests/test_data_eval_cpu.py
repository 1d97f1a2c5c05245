"""CPU checks of the oracle for the callers either side of the hot path (window builder, evaluation metrics)."""
import numpy as np
import torch

import oracle
from conftest import load_golden


def test_window_batch_matches_the_reference_semantics():
    """ClimateDataset.__getitem__ (main_final.py:97-154): the window ENDS at idx, left-padded with zero frames."""
    n, c, h, w, T = 7, 2, 3, 4, 3
    inp = torch.arange(n * c * h * w, dtype=torch.float32).reshape(n, c, h, w) + 1.0
    out = -torch.arange(n * 1 * h * w, dtype=torch.float32).reshape(n, 1, h, w)
    x, y = oracle.data_ref.window_batch(inp, out, [0, 1, 2, 6], T)
    assert x.shape == (4, T, c, h, w) and y.shape == (4, 1, h, w)
    assert torch.equal(x[0, 0], torch.zeros(c, h, w)) and torch.equal(x[0, 1], torch.zeros(c, h, w))
    assert torch.equal(x[0, 2], inp[0]) and torch.equal(y[0], out[0])
    assert torch.equal(x[1, 0], torch.zeros(c, h, w)) and torch.equal(x[1, 1], inp[0]) and torch.equal(x[1, 2], inp[1])
    assert torch.equal(x[2], inp[0:3]) and torch.equal(x[3], inp[4:7]) and torch.equal(y[3], out[6])


def test_climate_metrics_match_the_kaggle_metric():
    """oracle.data_ref.climate_metrics + climate_amd.evaluation.kaggle_score == _climate_kaggle_metric.score on the
    synthetic fields of the reference's own test (_test_kaggle_metric.py): the one function the reference pins."""
    from climate_amd.evaluation import kaggle_score
    g = load_golden("kaggle_metric.npz")
    lats = np.round(g["lats"].numpy() if hasattr(g["lats"], "numpy") else g["lats"], 2)      # the IDs carry %.2f latitudes
    pred = np.stack([np.asarray(g["tas_pred"]), np.asarray(g["pr_pred"])], axis=1)
    true = np.stack([np.asarray(g["tas_true"]), np.asarray(g["pr_true"])], axis=1)
    m = oracle.data_ref.climate_metrics(pred, true, lats)
    s = kaggle_score({"tas": m[0], "pr": m[1]})
    assert abs(s - float(g["score"])) < 1e-9 * float(g["score"])


def test_inverse_transform_is_the_inverse_of_normalize():
    stats = {0: {"method": "zscore", "params": {"mean": 280.0, "std": 12.0}},
             1: {"method": "log1p", "params": {"mean": 0.7, "std": 0.5}}}
    rng = np.random.default_rng(0)
    phys = np.stack([280 + 12 * rng.standard_normal((5, 4, 6)), np.abs(rng.standard_normal((5, 4, 6))) * 3], axis=1)
    norm = np.stack([(phys[:, 0] - 280.0) / 12.0, (np.log1p(phys[:, 1]) - 0.7) / 0.5], axis=1)
    back = oracle.data_ref.inverse_transform(norm, stats)
    assert np.allclose(back, phys, rtol=1e-12, atol=1e-12)


def test_denorm_params_surface():
    from climate_amd.evaluation import denorm_params
    p = denorm_params({0: {"method": "zscore", "params": {"mean": 1.0, "std": 2.0}},
                       1: {"method": "minimax", "params": {"min_val": 0.0, "max_val": 550.0}}}, 3)
    assert p.tolist() == [[1.0, 1.0, 2.0, 1.0], [2.0, 0.0, 550.0, 1.0], [0.0, 0.0, 1.0, 1.0]]
    try:
        denorm_params({0: {"method": "nope", "params": {}}}, 1)
        raise AssertionError("unknown method accepted")
    except ValueError as e:
        assert "Unknown inverse method" in str(e)
