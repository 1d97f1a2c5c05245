"""Data-parallel gradient exchange: one process per GPU, one flat fp32 all-reduce per step over RCCL/xGMI.

The reference has no explicit collective: with ``devices > 1`` Lightning wraps the module in DDP (main_final.py:768),
i.e. a bucketed NCCL all-reduce-mean of the gradients plus a parameter broadcast from rank 0.  Here the 73 trainable
tensors live in ONE flat buffer (model.flatten_parameters_), so the exchange is a single ``all_reduce(SUM)`` of
14.6 MB (base 32) / 58.5 MB (base 64); the 1/world scale is folded into the fused Adam (``grad_scale``).  The
unused ``post_conv.*`` parameters sit after the trainable prefix and are never communicated (stock DDP would raise
on them).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests of this logic.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> tuple:
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0) -> None:
    """DDP's initial parameter sync: every rank adopts rank ``src``'s flat parameter buffer."""
    if world_size() > 1:
        dist.broadcast(flat_params, src=src)


def allreduce_gradients(flat_grads: torch.Tensor, async_op: bool = False):
    """SUM all-reduce of the flat gradient prefix; returns the 1/world factor the optimizer must apply."""
    w = world_size()
    work = None
    if w > 1:
        work = dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op)
    return 1.0 / w, work


def shard_batch(n_global: int, rank: int, world: int) -> slice:
    """Contiguous, equal per-rank share of a global batch (Lightning semantics: batch_size is per device)."""
    if n_global % world:
        raise ValueError(f"global batch {n_global} is not divisible by world size {world}")
    per = n_global // world
    return slice(rank * per, (rank + 1) * per)


def mean_scalar(t: torch.Tensor) -> torch.Tensor:
    """self.log(..., sync_dist=True) equivalent for a logged scalar."""
    if world_size() > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
    return t
