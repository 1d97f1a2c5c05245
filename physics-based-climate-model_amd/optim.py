"""``HipAdam``: torch.optim.Adam semantics (coupled L2, bias correction) on the hand-written fused kernel.

Reference: ``optim.Adam(self.parameters(), lr=..., weight_decay=...)`` in ``configure_optimizers``
(main_final.py:737-747).  Parameters without gradients (``post_conv.*``) are skipped, exactly as torch does.
"""
import torch

from . import ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros(p.numel() + 4, device=p.device, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros(p.numel() + 4, device=p.device, dtype=torch.float32)
                st["step"] += 1
                flat_p = p.data.view(-1)
                g = p.grad.contiguous().view(-1)
                if flat_p.data_ptr() % 16 or g.data_ptr() % 16:
                    # unaligned view (e.g. a slice of someone else's buffer): go through an aligned staging copy
                    tmp_p, tmp_g = flat_p.clone(), g.clone()
                    ops.adam_step(tmp_p, tmp_g, st["exp_avg"][:p.numel()], st["exp_avg_sq"][:p.numel()], st["step"],
                                  group["lr"], b1, b2, group["eps"], group["weight_decay"])
                    flat_p.copy_(tmp_p)
                else:
                    ops.adam_step(flat_p, g, st["exp_avg"][:p.numel()], st["exp_avg_sq"][:p.numel()], st["step"],
                                  group["lr"], b1, b2, group["eps"], group["weight_decay"])
        return loss
