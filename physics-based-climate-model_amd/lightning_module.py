"""``ClimateEmulationModule``: the train-side surface of the reference's LightningModule.

Mirrors main_final.py:538-561 (``__init__(model, learning_rate, weight_decay)``, ``forward``, ``training_step``) and
main_final.py:737-747 (``configure_optimizers``).  When ``lightning`` is importable it subclasses
``lightning.pytorch.LightningModule`` so ``pl.Trainer.fit`` drives it unchanged; otherwise (this image) it is a
plain ``nn.Module`` with the same methods, driven by ``climate_amd.trainer``.  Validation / test / Kaggle export
are outside the hot-path scope.
"""
import types

import torch
import torch.nn as nn

from . import ops
from .optim import HipAdam

try:  # pragma: no cover - lightning is not in this image
    import lightning.pytorch as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError here
    _Base = nn.Module
    HAVE_LIGHTNING = False


class _MSEFunction(torch.autograd.Function):
    """nn.MSELoss() (main_final.py:544) on the hand-written reduction kernel."""

    @staticmethod
    def forward(ctx, pred, target):
        loss, dpred = ops.mse_loss(pred.contiguous(), target.contiguous(), want_grad=True)
        ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dpred,) = ctx.saved_tensors
        return dpred * dloss, None


def mse_loss(pred, target):
    if pred.shape != target.shape:
        raise RuntimeError(f"MSE shape mismatch: {tuple(pred.shape)} vs {tuple(target.shape)}")
    return _MSEFunction.apply(pred, target)


class ClimateEmulationModule(_Base):
    def __init__(self, model: nn.Module, learning_rate: float, weight_decay: float = 0.0):
        super().__init__()
        self.model = model
        if HAVE_LIGHTNING:  # pragma: no cover
            self.save_hyperparameters(ignore=["model"])
        else:
            self.hparams = types.SimpleNamespace(learning_rate=learning_rate, weight_decay=weight_decay)
        self.criterion = mse_loss
        self.normalizer = None
        self.logged = {}

    def forward(self, x):
        return self.model(x)

    def training_step(self, batch, batch_idx):
        x, y_true_norm = batch
        y_pred_norm = self(x)
        loss = self.criterion(y_pred_norm, y_true_norm)
        if HAVE_LIGHTNING:  # pragma: no cover
            self.log("train/loss", loss, prog_bar=True, batch_size=x.size(0))
        else:
            self.logged["train/loss"] = loss.detach()
        return loss

    def configure_optimizers(self):
        hp = self.hparams
        lr = hp.learning_rate if hasattr(hp, "learning_rate") else hp["learning_rate"]
        wd = getattr(hp, "weight_decay", 0.0) if not isinstance(hp, dict) else hp.get("weight_decay", 0.0)
        return HipAdam(self.parameters(), lr=lr, weight_decay=wd or 0.0)
