"""N>1 path on CPU (gloo, world size 2): the gradient bucketing/all-reduce wrapper that replaces the
reference's nn.DataParallel (trainer.py:228-229), and the sharding identity it relies on:
full-batch gradient == mean of per-rank shard gradients because every loss term is a batch mean."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _worker_gradsync(rank, world, port, ret):
    _init(rank, world, port)
    from hdf_rt import _lib
    from hdf_rt.parallel import GradSync, bucket_bounds
    from models.HDenseFormer import HDenseFormer_16
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT (the trainer seeds after init)
    net = HDenseFormer_16(in_channels=2, n_cls=3, image_size=(32, 32, 32), transformer_depth=4)
    before = net.flat_parameters().clone()
    sync = GradSync(net)
    after = net.flat_parameters()
    gathered = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    ok_bcast = all(torch.equal(gathered[0], g) for g in gathered) and (rank == 0) == bool(torch.equal(before, after))
    # views still alias the flat buffer after the broadcast
    p0 = next(net.parameters())
    ok_alias = p0.data_ptr() == after.data_ptr()
    # three-bucket all-reduce (one per backward stage): every element ends up as the mean over ranks
    g = net.flat_grads()
    g.copy_(torch.arange(g.numel(), dtype=torch.float32) % 97 + 1000.0 * rank)
    expect = torch.arange(g.numel(), dtype=torch.float32) % 97 + 1000.0 * (world - 1) / 2.0
    (lo1, hi1), (lo2, hi2), (lo3, hi3) = bucket_bounds(net._plan(_lib.F32).table)
    sync(1)
    part_done = torch.allclose(g[lo1:hi1], expect[lo1:hi1]) and not torch.allclose(g[lo2:hi2], expect[lo2:hi2]) \
        and not torch.allclose(g[lo3:hi3], expect[lo3:hi3])
    sync(2)
    part_done = part_done and torch.allclose(g[lo2:hi2], expect[lo2:hi2]) and not torch.allclose(g[lo3:hi3], expect[lo3:hi3])
    sync(3)
    sync.wait()
    ok_mean = torch.allclose(g, expect)
    covers = lo3 == 0 and hi3 == lo2 and hi2 == lo1 and hi1 >= g.numel() and lo3 < hi3 < hi2 < hi1
    ret[rank] = bool(ok_bcast and ok_alias and part_done and ok_mean and covers)
    dist.destroy_process_group()


def _worker_shard_identity(rank, world, port, ret):
    _init(rank, world, port)
    from oracle import detgen
    from oracle import hdf_oracle as orc
    cfg = (2, 3, 16, (32, 32, 32), 4)
    sd = orc.det_model(*cfg)
    x = torch.from_numpy(detgen.det_input(world, cfg[0], cfg[3], tag="ddp"))
    oh = torch.from_numpy(detgen.one_hot(detgen.det_labels(world, cfg[1], cfg[3], tag="ddp"), cfg[1]))
    tr = orc.OracleTrainer(sd)
    tr.loss_and_grads(x[rank:rank + 1], oh[rank:rank + 1])                 # this rank's shard (per-rank batch 1)
    flat = torch.cat([v.grad.flatten() for v in tr.sd.values()])
    dist.all_reduce(flat)
    flat /= world
    if rank == 0:
        full = orc.OracleTrainer(sd)
        full.loss_and_grads(x, oh)                                       # the single-process global batch
        ref = torch.cat([v.grad.flatten() for v in full.sd.values()])
        ret[0] = float((flat - ref).norm() / ref.norm())
    dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(fn, args=(world, _free_port(), ret), nprocs=world, join=True)
    return dict(ret)


def test_gradsync_broadcast_and_bucketed_allreduce_world2():
    ret = _run(_worker_gradsync)
    assert ret == {0: True, 1: True}


def test_sharded_gradients_equal_full_batch_gradients_world2():
    ret = _run(_worker_shard_identity)
    assert ret[0] < 5e-3          # fp32 noise floor of the reference's own gradients
