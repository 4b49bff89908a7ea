import sys, os, time, gc, json, subprocess
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
size = int(os.environ.get("SIZE", "256"))
if os.environ.get("CHILD") == "1":
    r = subprocess.run([sys.executable, "/root/repo/bench.py", "--cpu-baseline-only", "--size", str(size)], capture_output=True, text=True)
    print("child rc", r.returncode)
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx, size); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8, S=size), "cuda")
hp = ops.chain_stream(); hp.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(hp)
if os.environ.get("RESERVE_GB"):
    x = torch.empty(int(float(os.environ["RESERVE_GB"]) * (1 << 30)), dtype=torch.uint8, device="cuda"); del x
for _ in range(5): tr.train_step(batch, next_batch=batch)
gc.collect(); gc.freeze()
torch.cuda.synchronize()
ms = torch.cuda.memory_stats
rows = []
marks = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
marks[0].record()
for i in range(20):
    h = time.perf_counter()
    s0 = ms()
    tr.train_step(batch, next_batch=batch)
    marks[i + 1].record()
    s1 = ms()
    rows.append(((time.perf_counter() - h) * 1e3, (s1["reserved_bytes.all.current"] - s0["reserved_bytes.all.current"]) / 2**20,
                 s1["segment.all.allocated"] - s0["segment.all.allocated"], s1.get("num_alloc_retries", 0), gc.get_count()))
torch.cuda.synchronize()
step = [marks[i].elapsed_time(marks[i + 1]) for i in range(20)]
for i, r in enumerate(rows):
    print("step %2d gpu %6.1f ms host %6.1f ms  reserved +%8.1f MiB  new segments %d  retries %d gc %s" % (i, step[i], r[0], r[1], r[2], r[3], r[4]))
print("reserved total %.1f GiB" % (ms()["reserved_bytes.all.current"] / 2**30))
