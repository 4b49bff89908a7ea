"""Per-step GPU time, host enqueue time, allocator segments and garbage collections of the bf16 step (round 5: 85-107 ms steps every
4-6 steps after the bf16-storage change).  usage: python profiles/experiments/r5_spikes.py [storage 0|1] [steps]"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jafpro_amd import ops, synth  # noqa: E402
from jafpro_amd.step import Stage4Trainer, _to_dev  # noqa: E402

storage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ops.set_precision("bf16")
ops.set_bf16_storage(bool(storage))
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx)
M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
hp = ops.chain_stream()
hp.wait_stream(torch.cuda.current_stream())
torch.cuda.set_stream(hp)
warm = int(os.environ.get("R5_WARM", "5"))
spare = float(os.environ.get("R5_SPARE_GB", "0"))
for i in range(warm):
    tr.train_step(batch, next_batch=batch)
    if i == 0 and spare > 0:
        t = torch.empty(int(spare * 2 ** 30), dtype=torch.uint8, device="cuda")
        del t
print("after warm-up: reserved %.1f GB, peak allocated %.1f GB" % (torch.cuda.memory_reserved() / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30))
gc.collect()
gc.freeze()
torch.cuda.synchronize()
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get("generation"), time.perf_counter())) if phase == "stop" else None)
marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
marks[0].record()
rows = []
for i in range(steps):
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    n_gc = len(gcs)
    tr.train_step(batch, next_batch=batch)
    marks[i + 1].record()
    t1 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    rows.append(((t1 - t0) * 1e3, st1["num_device_alloc"] - st0["num_device_alloc"], st1["num_device_free"] - st0["num_device_free"],
                 st1["num_alloc_retries"] - st0["num_alloc_retries"], (st1["reserved_bytes.all.current"] - st0["reserved_bytes.all.current"]) / 2 ** 20,
                 len(gcs) - n_gc, st1["reserved_bytes.all.current"] / 2 ** 30))
torch.cuda.synchronize()
print("storage", storage)
for i, r in enumerate(rows):
    print("step %2d gpu %7.2f ms host %7.2f ms  hipMalloc %d hipFree %d retries %d reserved %+8.1f MB (%.1f GB) gc %d"
          % (i, marks[i].elapsed_time(marks[i + 1]), r[0], r[1], r[2], r[3], r[4], r[6], r[5]))
