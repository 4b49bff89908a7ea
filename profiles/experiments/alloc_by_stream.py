"""Which stream's pool holds the caching allocator's reserve?  After 40 train steps: per stream the bytes of its segments, of which
active, plus the peak of allocated bytes; and the same after torch.cuda.empty_cache() at the step boundary."""
import sys, torch, collections
sys.path.insert(0, ".")
import bench
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
def report(tag):
    torch.cuda.synchronize()
    snap = torch.cuda.memory_snapshot()
    per = collections.defaultdict(lambda: [0, 0, 0, 0])
    for seg in snap:
        p = per[seg["stream"]]
        p[0] += seg["total_size"]; p[1] += seg["active_size"]; p[2] += 1
        p[3] = max(p[3], max((b["size"] for b in seg["blocks"] if b["state"] == "inactive"), default=0))
    print("== %s: allocated %.1f GB, reserved %.1f GB, peak allocated %.1f GB" % (tag, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9))
    for st, (tot, act, n, big) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("   stream %-14s segments %4d  reserved %7.2f GB  active now %7.2f GB  largest free block %6.2f GB" % (st, n, tot / 1e9, act / 1e9, big / 1e9))
for i in range(40):
    tr.train_step(batch, next_batch=batch)
report("after 40 steps")
torch.cuda.synchronize(); torch.cuda.empty_cache()
report("after empty_cache at a step boundary")
torch.cuda.reset_peak_memory_stats()
import time
ts = []
for i in range(20):
    t0 = time.time(); tr.train_step(batch, next_batch=batch); ts.append((time.time() - t0) * 1e3)
report("20 steps later")
print("host ms per step after empty_cache:", " ".join("%.0f" % t for t in ts))
