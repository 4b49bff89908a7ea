import sys, gc, time, torch
sys.path.insert(0, ".")
import bench
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
for _ in range(5): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
print("objects tracked:", len(gc.get_objects()), "counts", gc.get_count(), "thresholds", gc.get_threshold())
log = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: log.append((info["generation"], (time.perf_counter() - cb.t) * 1e3, info["collected"]))
gc.callbacks.append(cb)
t = time.perf_counter(); n = gc.collect(); print("full collect: %.1f ms, %d collected" % ((time.perf_counter() - t) * 1e3, n))
log.clear()
for i in range(40): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
by = {}
for g, ms, c in log: by.setdefault(g, []).append(ms)
for g, v in sorted(by.items()): print("gen %d: %d collections in 40 steps, max %.1f ms, total %.1f ms" % (g, len(v), max(v), sum(v)))
gc.freeze()
log.clear()
for i in range(40): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
by = {}
for g, ms, c in log: by.setdefault(g, []).append(ms)
for g, v in sorted(by.items()): print("after gc.freeze(): gen %d: %d collections, max %.1f ms, total %.1f ms" % (g, len(v), max(v), sum(v)))
