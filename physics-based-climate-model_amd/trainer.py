"""Fused training step for the hot path: zero grads -> forward -> MSE -> backward -> [RCCL all-reduce] -> Adam.

Equivalent to one iteration of Lightning's automatic optimisation around ``training_step``
(main_final.py:556-561, 737-747) with ``optimizer.zero_grad(); loss.backward(); optimizer.step()``, but scheduled
directly on the engine: flat parameter / gradient / moment buffers, one fused Adam launch, and (optionally) the whole
device-side step recorded once into a hipGraph and replayed, which removes ~300 launches' worth of host time.
"""
from typing import Optional

import torch

from . import ddp, engine, ops
from ._lib import check, lib


class HotPathTrainer:
    def __init__(self, model, lr: float = 5e-4, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 use_graph: bool = True, distributed: Optional[bool] = None):
        self.model = model
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("HotPathTrainer needs the model on the GPU")
        self.device = dev
        self.flat = model._flat if model._flat is not None else model.flatten_parameters_()
        self.nt = model.n_flat_trainable
        self.grad = torch.zeros(self.nt, device=dev, dtype=torch.float32)
        self.m = torch.zeros(self.nt, device=dev, dtype=torch.float32)
        self.v = torch.zeros(self.nt, device=dev, dtype=torch.float32)
        self.adam_state = torch.zeros(4, device=dev, dtype=torch.float32)   # device-side step counter + corrections
        self.loss = torch.zeros(1, device=dev, dtype=torch.float32)
        self.world = ddp.world_size() if distributed is None else (ddp.world_size() if distributed else 1)
        if self.world > 1:
            ddp.broadcast_parameters(self.flat)
        self.use_graph = use_graph
        self._graphs = {}
        self._static = {}
        self.steps = 0

    # ------------------------------------------------------------------ pieces
    def _fwd_bwd(self, x, y):
        p = self.model._param_dict()
        g = self.model._views(self.grad)
        check(lib.cm_zero(self.grad.data_ptr(), self.nt * 4, torch.cuda.current_stream().cuda_stream), "zero")
        pk = engine.get_plan(p, None, False).pack()
        pred, sv = engine.forward(p, pk, x, save=True)
        check(lib.cm_mse_loss(pred.data_ptr(), y.data_ptr(), self.loss.data_ptr(), pred.data_ptr(), pred.numel(),
                              torch.cuda.current_stream().cuda_stream), "mse")       # dpred overwrites pred in place
        engine.backward(p, pk, g, sv, pred, need_dx=False)

    def _adam(self):
        b1, b2 = self.betas
        check(lib.cm_adam_step_dev(self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                   self.nt, self.adam_state.data_ptr(), self.lr, b1, b2, self.eps, self.wd,
                                   1.0 / self.world, torch.cuda.current_stream().cuda_stream), "adam")

    def _eager_step(self, x, y):
        self._fwd_bwd(x, y)
        if self.world > 1:
            ddp.allreduce_gradients(self.grad)
        self._adam()

    # ------------------------------------------------------------------ public
    def step(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One optimisation step on batch (x [B,T,C,H,W], y [B,out,H,W]); returns the (device) loss tensor [1]."""
        if not (x.is_cuda and y.is_cuda):
            raise RuntimeError("batch must be on the GPU")
        x = x.contiguous()
        y = y.contiguous()
        self.steps += 1
        if not self.use_graph:
            self._eager_step(x, y)
            return self.loss
        key = (tuple(x.shape), tuple(y.shape))
        if key not in self._graphs:
            self._capture(key, x, y)
        sx, sy = self._static[key]
        sx.copy_(x, non_blocking=True)
        sy.copy_(y, non_blocking=True)
        g1, g2 = self._graphs[key]
        g1.replay()
        if self.world > 1:
            ddp.allreduce_gradients(self.grad)
        g2.replay()
        return self.loss

    def _capture(self, key, x, y):
        """Record {fwd, loss, bwd} and {Adam} as two hipGraphs; the RCCL all-reduce runs between them."""
        sx, sy = x.clone(), y.clone()
        # warm-up on a side stream (lazy module loading, allocator pools) WITHOUT touching the optimizer state
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._fwd_bwd(sx, sy)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            self._fwd_bwd(sx, sy)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            self._adam()
        self._graphs[key] = (g1, g2)
        self._static[key] = (sx, sy)
