"""``HipAdam``: torch.optim.Adam semantics (coupled L2, bias correction) on the hand-written fused kernel.

Reference: ``optim.Adam(self.parameters(), lr=..., weight_decay=...)`` in ``configure_optimizers``
(main_final.py:737-747).  Parameters without gradients (``post_conv.*``) are skipped, exactly as torch does.

The optimizer STATE has torch.optim.Adam's layout -- per parameter ``{"step": 0-dim float32 tensor, "exp_avg":
zeros_like(p), "exp_avg_sq": zeros_like(p)}`` -- so ``state_dict()`` / ``load_state_dict()`` interoperate with a
checkpoint written by the reference (Lightning restores the optimizer state on ``ckpt_path=`` resume) in both
directions.  ``flat_adam_state_dict`` / ``load_flat_adam_state`` map the fused trainer's flat moment buffers to the
same format.
"""
from typing import Dict, List, Tuple

import torch

from . import ops


def _step_tensor(value: float = 0.0) -> torch.Tensor:
    # torch.optim.Adam (non-capturable, non-fused) keeps `step` as a 0-dim float32 CPU tensor
    return torch.tensor(float(value), dtype=torch.float32)


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        # (the extra keys are torch.optim.Adam's own defaults: a reference checkpoint's param_groups carry them, and
        #  a checkpoint written here loads into torch.optim.Adam without missing-key surprises)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("decoupled_weight_decay"):
                raise RuntimeError("HipAdam implements plain Adam (amsgrad / maximize / decoupled decay off), as the "
                                   "reference uses it")
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = _step_tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not torch.is_tensor(st["step"]):         # (a state written by an older build kept a python int)
                    st["step"] = _step_tensor(st["step"])
                st["step"] += 1
                t = int(round(float(st["step"])))
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m.device != p.device:                     # loaded from a CPU checkpoint
                    m = st["exp_avg"] = m.to(p.device)
                    v = st["exp_avg_sq"] = v.to(p.device)
                if not (m.is_contiguous() and v.is_contiguous()):
                    m = st["exp_avg"] = m.contiguous()
                    v = st["exp_avg_sq"] = v.contiguous()
                flat_p = p.data.view(-1)
                g = p.grad.contiguous().view(-1)
                fm, fv = m.view(-1), v.view(-1)
                if (flat_p.data_ptr() | g.data_ptr() | fm.data_ptr() | fv.data_ptr()) % 16:
                    # unaligned view (e.g. a slice of someone else's buffer): go through aligned staging copies
                    tp, tg, tm, tv = flat_p.clone(), g.clone(), fm.clone(), fv.clone()
                    ops.adam_step(tp, tg, tm, tv, t, group["lr"], b1, b2, group["eps"], group["weight_decay"])
                    flat_p.copy_(tp)
                    fm.copy_(tm)
                    fv.copy_(tv)
                else:
                    ops.adam_step(flat_p, g, fm, fv, t, group["lr"], b1, b2, group["eps"], group["weight_decay"])
        return loss


# ------------------------------------------------------------------------------------------------- flat <-> torch
def flat_adam_state_dict(layout: Dict[str, Tuple[int, int, tuple]], param_names: List[str], trainable: List[str],
                         m: torch.Tensor, v: torch.Tensor, step: int, lr: float, betas, eps: float,
                         weight_decay: float) -> dict:
    """torch.optim.Adam-format ``state_dict`` from the fused trainer's flat moment buffers.

    ``layout``: name -> (offset, numel, shape) inside the flat buffers; ``param_names``: ``model.named_parameters()``
    order (= the optimizer's parameter indices, 75 entries); ``trainable``: names that carry gradients.  Parameters
    outside ``trainable`` (``post_conv.*``) get no state entry, exactly like torch (their grad is None).  Before the
    first step the state is empty, as in torch."""
    state = {}
    if step > 0:
        tr = set(trainable)
        for idx, name in enumerate(param_names):
            if name not in tr:
                continue
            o, k, shape = layout[name]
            state[idx] = {"step": _step_tensor(step),
                          "exp_avg": m[o:o + k].detach().reshape(shape).clone(),
                          "exp_avg_sq": v[o:o + k].detach().reshape(shape).clone()}
    group = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False,
                 params=list(range(len(param_names))))
    return {"state": state, "param_groups": [group]}


def load_flat_adam_state(sd: dict, layout: Dict[str, Tuple[int, int, tuple]], param_names: List[str],
                         m: torch.Tensor, v: torch.Tensor) -> int:
    """Inverse of flat_adam_state_dict: fills the flat moment buffers from a torch.optim.Adam ``state_dict`` (ours or
    the reference's) and returns the common step count.  Raises if the entries disagree about the step."""
    groups = sd["param_groups"]
    order = [i for g in groups for i in g["params"]]
    if len(order) != len(param_names):
        raise ValueError(f"optimizer state has {len(order)} parameters, the model {len(param_names)}")
    m.zero_()
    v.zero_()
    steps = set()
    for pos, idx in enumerate(order):
        st = sd["state"].get(idx)
        if not st:
            continue
        name = param_names[pos]
        if name not in layout:
            raise ValueError(f"optimizer state for {name}, which the fused step does not train")
        o, k, shape = layout[name]
        if o + k > m.numel():
            raise ValueError(f"optimizer state for {name}, which lies outside the trainable prefix")
        ea, es = st["exp_avg"], st["exp_avg_sq"]
        if tuple(ea.shape) != tuple(shape):
            raise ValueError(f"exp_avg shape {tuple(ea.shape)} != parameter shape {tuple(shape)} for {name}")
        m[o:o + k].copy_(ea.reshape(-1))
        v[o:o + k].copy_(es.reshape(-1))
        steps.add(int(round(float(st["step"]))))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state the fused step can resume")
    return steps.pop() if steps else 0
