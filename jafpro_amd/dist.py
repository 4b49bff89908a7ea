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

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def limit_hw_queues(n: int = 2) -> bool:
    """Call BEFORE the process touches the GPU when it will also run RCCL.  HIP gives a process up to GPU_MAX_HW_QUEUES (default 4)
    hardware queues per stream priority; the step's side streams use the 4 normal-priority ones plus 1 high-priority queue for the
    dependent chain, and RCCL brings one more of its own.  Six queues oversubscribe the hardware queues of an MI355X and the step
    runs 50 % slower (measured with a one-rank group, `profiles/experiments/round4_x4.log`: 54 -> 81 ms; the same 79-80 ms that
    GPU_MAX_HW_QUEUES=5 alone produces); three normal-priority queues bring it back to 55.9-56.2 ms, two to 57.1-57.4.  Two is the default:
    it leaves room for a second queue of RCCL's at larger world sizes, which a 1-GPU box cannot show, and one queue too many costs
    25 ms where one too few costs 1 (export GPU_MAX_HW_QUEUES=3 where the faster setting is known to hold).  Returns False (and changes nothing) when the variable is already set by the user or HIP is already initialised."""
    import os
    import warnings
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return False
    if torch.cuda.is_initialized():
        warnings.warn("jafpro_amd.dist.limit_hw_queues() called after HIP was initialised: GPU_MAX_HW_QUEUES can no longer be set "
                      "for this process and a multi-rank step will run ~50 %% slower; call it before the first GPU call "
                      "(or export GPU_MAX_HW_QUEUES=%d in the launcher)" % hw_queues_default(n), RuntimeWarning, stacklevel=2)
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(hw_queues_default(n))
    return True


def hw_queues_default(n: int = 2) -> int:
    """JAF_HW_QUEUES overrides the size of the normal-priority queue pool `limit_hw_queues` asks for (2 is the safe default; 3 was 1 ms
    faster with a one-rank group, whether it holds at N = 8 can only be measured on an 8-GPU node)."""
    import os
    try:
        return max(1, int(os.environ.get("JAF_HW_QUEUES", n)))
    except ValueError:
        return int(n)


def warn_if_hw_queues_unset(reducer) -> None:
    """A trainer with an active RCCL reducer in a process whose launcher never called `limit_hw_queues`: say so once (the step is
    correct either way, but six hardware queues cost 50 % of it: profiles/experiments/round4_x4.log)."""
    import os
    import warnings
    if reducer is None or not getattr(reducer, "active", False) or "GPU_MAX_HW_QUEUES" in os.environ:
        return
    try:
        if dist.get_backend(reducer.group) != "nccl":
            return
    except Exception:
        return
    warnings.warn("multi-rank trainer over RCCL without GPU_MAX_HW_QUEUES: call jafpro_amd.dist.limit_hw_queues() before the first "
                  "GPU call of every rank (or export GPU_MAX_HW_QUEUES=2)", RuntimeWarning, stacklevel=3)


class GradReducer:
    def __init__(self, group=None, bucket_bytes: int = 64 << 20, skip_single: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.active = not (skip_single and self.world == 1)   # a 1-rank group still exercises the path in tests
        # host-side metadata (per-rank face counts) travels over gloo: an RCCL group only moves device tensors,
        # and reading those back would put a device sync into every step.  Every rank constructs the reducer,
        # so every rank makes this collective new_group call.
        self._host_group = group
        if self.active and dist.get_backend(group) != "gloo":
            self._host_group = dist.new_group(backend="gloo")

    def host_allgather_int(self, value: int) -> List[int]:
        t = torch.tensor([int(value)], dtype=torch.int64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self._host_group)
        return [int(o.item()) for o in out]

    def begin(self, buffers: Sequence[torch.Tensor]) -> List[Tuple[object, torch.Tensor]]:
        """Issues the sum over ranks of every flat buffer as <= bucket_bytes asynchronous messages
        (they pipeline on the xGMI links) and returns the handles for `finish`.  On RCCL the messages
        wait for the work already enqueued on the current stream and then run beside whatever is
        enqueued next -- e.g. the rest of the backward pass (`BackwardOverlap`)."""
        if not self.active:
            return []
        works = []
        for buf in buffers:
            flat = buf.view(-1)
            for off in range(0, flat.numel(), self.bucket_elems):
                chunk = flat[off:off + self.bucket_elems]
                works.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk))
        return works

    def finish(self, works) -> None:
        """Waits for the messages of `begin` (the current stream waits, not the host) and turns the
        sums into means."""
        inv = 1.0 / self.world
        for w, chunk in works:
            w.wait()
            if chunk.is_cuda:
                from . import ops
                ops.axpby(inv, chunk, 0.0, chunk)
            else:
                chunk.mul_(inv)

    def all_reduce_mean(self, buffers: Sequence[torch.Tensor]):
        """In-place mean over ranks of every flat buffer, all messages issued before the first wait."""
        self.finish(self.begin(buffers))

    def broadcast(self, buffers: Sequence[torch.Tensor], src: int = 0):
        for buf in buffers:
            dist.broadcast(buf, src=src, group=self.group)
        if any(b.is_cuda for b in buffers):
            from . import ops
            ops.invalidate_packed_weights()

    def host_broadcast_ints(self, values: Sequence[int], src: int = 0) -> List[int]:
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        dist.broadcast(t, src=src, group=self._host_group)
        return [int(v) for v in t]

    @staticmethod
    def checksum(buffers: Sequence[torch.Tensor]) -> torch.Tensor:
        """int64 [len(buffers)]: the wrapping sum of every buffer's 32-bit patterns.  Integer addition is exact and order-free, so
        two ranks holding bit-identical buffers get the same number whatever their reduction order; a single differing bit in
        one element changes it."""
        return torch.stack([b.reshape(-1).view(torch.int32).sum(dtype=torch.int64) for b in buffers])

    def consistent(self, buffers: Sequence[torch.Tensor]) -> List[bool]:
        """Per buffer: do all ranks hold the same bits?  One small all-gather + ONE device read-back: not for every step
        (Stage4Trainer calls it every `check_every` steps)."""
        if not self.active:
            return [True] * len(buffers)
        mine = self.checksum(buffers)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        allv = torch.stack(out).cpu()
        return [bool((allv[:, i] == allv[0, i]).all()) for i in range(len(buffers))]


class BackwardOverlap:
    """Starts a module's gradient all-reduce from INSIDE the backward pass.

    `watch(x, name, buffers)`: x is the activation a module received as its input.  The hook on x fires
    when the gradient w.r.t. x has been produced, i.e. after every backward node between the loss and x
    has run; every parameter of the stage-4 modules sits on such a path (each convolution consumes an
    activation derived from the module input, and its weight gradient is written by the same backward
    node as its data gradient), so the module's flat gradient buffer is final and its messages can
    travel while the modules further upstream are still being differentiated.  `finish` starts whatever
    has not fired (modules whose input carries no gradient), waits for everything and takes the means.
    Message order is the reverse graph order on every rank, so ranks issue identical collective sequences.

    Gradients written on a side stream (ops.set_wgrad_stream): either `before_begin` joins that stream into the
    current one before a module's messages are issued (the dependent chain then stalls behind the weight-gradient
    backlog at every module boundary), or -- `issue_stream`: a callable returning that side stream AFTER making it
    wait for the current stream, or None -- the messages are issued FROM the side stream: RCCL's stream then waits
    for the module's weight gradients and for the chain up to the hook, and the chain itself does not wait at all.
    `finish(rest, each=f)` waits module by module in message order and calls f(name) as soon as that module's
    means are final (the trainer's optimiser step: it runs beside the messages still travelling)."""

    def __init__(self, reducer: GradReducer, before_begin=None, issue_stream=None):
        self.red = reducer
        self.before_begin = before_begin      # e.g. ops.join_wgrad_stream: gradients written on another stream
        self.issue_stream = issue_stream      # e.g. ops.wgrad_stream_after_current
        self.works = {}                       # name -> handles of its messages
        self.fired: List[str] = []
        self._hooks = []

    def _begin(self, name: str, buffers) -> None:
        self.fired.append(name)
        st = self.issue_stream() if self.issue_stream is not None else None
        if st is not None:
            with torch.cuda.stream(st):
                self.works[name] = self.red.begin(buffers)
            return
        if self.before_begin is not None:
            self.before_begin()
        self.works[name] = self.red.begin(buffers)

    def begin_now(self, name: str, buffers: Sequence[torch.Tensor]) -> None:
        """Issues `name`'s messages right away (a caller that knows by other means that these buffers are final, e.g. a parameter
        range of a module whose weight gradients have all been enqueued: ops.watch_wgrads)."""
        if name not in self.fired:
            self._begin(name, buffers)

    def watch(self, x: torch.Tensor, name: str, buffers: Sequence[torch.Tensor]) -> None:
        if x is None or not x.requires_grad:
            return

        def hook(_grad, name=name, buffers=buffers):
            if name not in self.fired:
                self._begin(name, buffers)
            return None

        self._hooks.append(x.register_hook(hook))

    def finish(self, rest: Sequence[Tuple[str, Sequence[torch.Tensor]]], each=None) -> None:
        for name, buffers in rest:
            if name not in self.fired:
                self._begin(name, buffers)
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for name in self.fired:               # message order = completion order on RCCL's stream
            self.red.finish(self.works.pop(name))
            if each is not None:
                each(name)
        self.works = {}


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
