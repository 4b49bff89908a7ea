"""N>1 path on CPU: two gloo ranks average flat gradient buffers exactly like the reference's
DataParallel reduction onto device 0 with equal shards."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jafpro_amd.dist import GradReducer
    red = GradReducer(bucket_bytes=4096)                 # force several buckets per buffer
    g = torch.Generator().manual_seed(100 + rank)
    bufs = [torch.randn(5000, generator=g), torch.randn(37, generator=g)]
    keep = [b.clone() for b in bufs]
    red.all_reduce_mean(bufs)
    # F10: the discriminator all-reduces ACCUMULATED grads g1, then g1+g2: reducing the running
    # buffer each time must equal accumulating reduced increments
    acc = torch.zeros(100)
    inc_sum = torch.zeros(100)
    for it in range(3):
        inc = torch.randn(100, generator=g)
        acc += inc
        red_acc = acc.clone()
        red.all_reduce_mean([red_acc])
        r_inc = inc.clone()
        red.all_reduce_mean([r_inc])
        inc_sum += r_inc
        assert torch.allclose(red_acc, inc_sum, atol=1e-6)
    p = [torch.full((10,), float(rank))]
    red.broadcast(p, src=0)
    out[rank] = (keep, bufs, p[0].clone())
    dist.destroy_process_group()


def test_two_rank_gradient_mean():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    k0, r0, p0 = out[0]
    k1, r1, p1 = out[1]
    for a, b, x, y in zip(k0, k1, r0, r1):
        assert torch.allclose(x, (a + b) / 2, atol=1e-6) and torch.equal(x, y)
    assert torch.equal(p0, torch.zeros(10)) and torch.equal(p1, torch.zeros(10))
