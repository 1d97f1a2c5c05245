"""CPU restatement of the callers either side of the hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

  window_batch        ClimateDataset.__getitem__ + default collation   main_final.py:97-154, 483-494
  inverse_transform   Normalizer.inverse_transform_output              src/utils_final.py:130-206
  climate_metrics     _evaluate_predictions' three metrics             main_final.py:576-632,
                                                                       src/utils_final.py:282-302,387-406

``main_final.py`` and ``src/utils_final.py`` are not importable here (lightning / hydra / xarray / dask are absent:
ordinary ModuleNotFoundError), so these are restatements; ``climate_metrics`` is pinned by the
reference's importable ``_climate_kaggle_metric.score`` on the synthetic data of its own test
(tests/golden/gen_golden.py -> kaggle_metric.npz), ``window_batch`` by the cases the reference's comments spell out.
"""
from typing import Dict, Sequence

import numpy as np
import torch


def window_batch(inputs: torch.Tensor, outputs: torch.Tensor, idxs: Sequence[int], seq_len: int):
    """[(input_seq [T,C,H,W], target [C_out,H,W]) for idx in idxs] stacked like the default collate_fn."""
    xs, ys = [], []
    pad = torch.zeros_like(inputs[0])                       # pad_tensor_template, main_final.py:76
    total = inputs.shape[0]
    for idx in idxs:
        parts = []
        for i in range(seq_len):
            cur = idx - seq_len + 1 + i                     # main_final.py:122
            parts.append(pad if (cur < 0 or cur >= total) else inputs[cur])
        xs.append(torch.stack(parts, dim=0))
        ys.append(outputs[idx])
    return torch.stack(xs), torch.stack(ys)


def inverse_transform(data_norm: np.ndarray, stats: Dict[int, dict]) -> np.ndarray:
    """data_norm [N, V, H, W] -> physical units, per variable."""
    out = []
    for v in range(data_norm.shape[1]):
        x = data_norm[:, v]
        cfg = stats.get(v)
        if cfg is None:
            out.append(x)
            continue
        m, p = cfg["method"], cfg.get("params", {})
        if m == "zscore":
            y = x * p["std"] + p["mean"]
        elif m == "minimax":
            y = x * (p["max_val"] - p["min_val"]) + p["min_val"]
        elif m == "log1p":
            y = np.expm1(x * p["std"] + p["mean"])
        elif m == "sqrt":
            y = (x * p["std"] + p["mean"]) ** 2
        elif m == "pow":
            y = (x * p["std"] + p["mean"]) ** (1.0 / p["lambda"])
        else:
            raise ValueError(f"Unknown inverse method '{m}' for var {v}.")
        out.append(y)
    return np.stack(out, axis=1)


def climate_metrics(pred: np.ndarray, true: np.ndarray, latitudes) -> np.ndarray:
    """pred, true [time, V, y, x] in physical units -> [V, 3]: monthly RMSE, time-mean RMSE, time-stddev MAE with
    cos(latitude) weights (weighted mean = sum(w * v) / sum(w) over the averaged dims)."""
    w = np.cos(np.deg2rad(np.asarray(latitudes, dtype=np.float64)))
    w = w / np.mean(w)
    res = []
    for v in range(pred.shape[1]):
        p, t = pred[:, v].astype(np.float64), true[:, v].astype(np.float64)
        wy = w[None, :, None]
        nyx = w.sum() * p.shape[2]
        monthly = np.sqrt(((p - t) ** 2 * wy).sum() / (nyx * p.shape[0]))
        tm = np.sqrt((((p.mean(0) - t.mean(0)) ** 2) * w[:, None]).sum() / nyx)
        ts = (np.abs(p.std(0) - t.std(0)) * w[:, None]).sum() / nyx
        res.append([monthly, tm, ts])
    return np.asarray(res)
