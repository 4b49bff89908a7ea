"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl")
all-reduce of the flat per-module gradient buffers over xGMI.

The reference's multi-GPU path is single-process nn.DataParallel around each module
(train/4.convLSTM_flowpro_interval.py:123-162): per-replica BatchNorm statistics, losses as
batch means, gradients summed onto device 0.  With equal shards that is exactly "average the
per-rank gradients", which is the only collective here; BatchNorm statistics stay
rank-local like the reference's.  Message sizes: generator 305.8 MB fp32 per step in four
per-module messages issued in reverse graph order, discriminator 3.94 MB x 3, face-D 1.35 MB.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, bucket_bytes: int = 64 << 20):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = max(1, bucket_bytes // 4)

    def all_reduce_mean(self, buffers: Sequence[torch.Tensor]):
        """In-place mean over ranks of every flat buffer.  Buffers are cut into <= bucket_bytes
        messages, all issued asynchronously before the first wait so that successive messages
        pipeline on the xGMI links."""
        if self.world == 1:
            return
        works = []
        for buf in buffers:
            flat = buf.view(-1)
            for off in range(0, flat.numel(), self.bucket_elems):
                chunk = flat[off:off + self.bucket_elems]
                works.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk))
        inv = 1.0 / self.world
        for w, chunk in works:
            w.wait()
            if chunk.is_cuda:
                from . import ops
                ops.axpby(inv, chunk, 0.0, chunk)
            else:
                chunk.mul_(inv)

    def broadcast(self, buffers: Sequence[torch.Tensor], src: int = 0):
        for buf in buffers:
            dist.broadcast(buf, src=src, group=self.group)
        if any(b.is_cuda for b in buffers):
            from . import ops
            ops.invalidate_packed_weights()


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Splits every batch-major array of a stage-4 batch into `world` equal shards (dim 0)."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        if n % world:
            raise ValueError("batch dim %d of %s not divisible by world size %d" % (n, k, world))
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out
