"""world_size-2 test of the data-parallel gradient exchange logic on the gloo backend (CPU).

Checks that SUM all-reduce + the 1/world factor folded into Adam reproduces the single-process update on the
concatenated batch, that the initial broadcast makes ranks identical, and that post_conv-style trailing entries of
the flat buffer are never communicated."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from climate_amd import ddp
    r, l, w = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and ddp.world_size() == world
    torch.manual_seed(100 + rank)                       # ranks start different ...
    nt, tail = 1000, 24                                 # trainable prefix + never-communicated tail
    flat = torch.randn(nt + tail)
    ddp.broadcast_parameters(flat[:nt])                 # ... and agree after the broadcast (tail untouched)
    torch.manual_seed(100)
    ref0 = torch.randn(nt + tail)
    assert torch.equal(flat[:nt], ref0[:nt])
    if rank != 0:
        assert not torch.equal(flat[nt:], ref0[nt:])
    # per-rank "gradients": mean-loss gradients of each rank's shard
    gen = torch.Generator().manual_seed(7)
    gfull = torch.randn(world, nt, generator=gen)       # rank i's local mean gradient
    g = gfull[rank].clone()
    scale, _ = ddp.allreduce_gradients(g)
    assert scale == 1.0 / world
    want = gfull.sum(0)
    assert torch.allclose(g, want, atol=1e-6)
    # Adam on (sum * 1/world) == Adam on the global-batch mean gradient
    p = flat[:nt].clone(); m = torch.zeros(nt); v = torch.zeros(nt)
    oracle.adam_reference_step(p, g * scale, m, v, 1, lr=5e-4)
    p_ref = ref0[:nt].clone(); m2 = torch.zeros(nt); v2 = torch.zeros(nt)
    oracle.adam_reference_step(p_ref, gfull.mean(0), m2, v2, 1, lr=5e-4)
    assert torch.allclose(p, p_ref, atol=1e-7)
    # two buckets, the first one asynchronous (what the trainer does beside the encoder's backward) == one exchange
    g2 = gfull[rank].clone()
    cut = 384
    _, work = ddp.allreduce_gradients(g2[cut:], async_op=True)
    ddp.allreduce_gradients(g2[:cut])
    work.wait()
    assert torch.allclose(g2, want, atol=1e-6)
    sl = ddp.shard_batch(64, rank, world)
    assert sl == slice(rank * 32, rank * 32 + 32)
    lg = ddp.mean_scalar(torch.tensor([float(rank)]))
    assert abs(lg.item() - (world - 1) / 2) < 1e-6
    ret[rank] = True
    dist.destroy_process_group()


def test_gradient_exchange_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_shard_batch_rejects_ragged():
    from climate_amd import ddp
    with pytest.raises(ValueError):
        ddp.shard_batch(33, 0, 2)
