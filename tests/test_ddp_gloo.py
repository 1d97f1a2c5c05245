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


def _worker8(rank, world, port, ret):
    """World 8 on the CPU: the per-rank batch shares of BASELINE configs 3 / 5 and the trainer's two-bucket exchange at the
    REAL bucket boundary of the model's flat gradient buffer (encoder prefix / ConvLSTM + decoder + head suffix), with the
    never-communicated post_conv tail behind it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from climate_amd import ddp
    from climate_amd.model import AttUNetConvLSTM
    ddp.init_from_env(backend="gloo")
    # config 3: global batch 256 -> 32 per rank; config 5: 128 -> 16
    for glob, per in ((256, 32), (128, 16)):
        sl = ddp.shard_batch(glob, rank, world)
        assert (sl.start, sl.stop) == (rank * per, (rank + 1) * per)
    torch.manual_seed(0)
    m = AttUNetConvLSTM(5, 2, 8, 3)                     # CPU construction: layout only, nothing is launched
    lay = m._build_layout()
    nt, bb = m.n_flat_trainable, m.bucket_boundary
    assert 0 < bb < nt and bb % 64 == 0
    assert all(o + k <= bb for n, (o, k, _) in lay.items() if n.startswith("enc"))
    assert all(o >= bb for n, (o, k, _) in lay.items() if not n.startswith("enc"))
    assert all(o >= nt for n, (o, k, _) in lay.items() if n.startswith("post_conv."))       # outside the exchange
    total = max(o + k for o, k, _ in lay.values())
    gen = torch.Generator().manual_seed(3)
    gfull = torch.randn(world, total, generator=gen)
    g = gfull[rank].clone()
    _, work = ddp.allreduce_gradients(g[bb:nt], async_op=True)      # suffix first, asynchronously (beside the encoder backward)
    ddp.allreduce_gradients(g[:bb])
    work.wait()
    assert torch.allclose(g[:nt], gfull[:, :nt].sum(0), atol=1e-5)
    assert torch.equal(g[nt:], gfull[rank, nt:])                    # the post_conv tail never travels
    ret[rank] = True
    dist.destroy_process_group()


def test_gradient_exchange_world8_buckets():
    world = 8
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))
