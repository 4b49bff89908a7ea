import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
import bench
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
for _ in range(3): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
evs = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((name, e))
tr.phase_mark = mark
tot = {}
for it in range(5):
    evs.clear(); mark("start")
    tr.train_step(batch, next_batch=batch)
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(evs[:-1], evs[1:]):
        tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1) / 5
for k, v in tot.items(): print("%-28s %7.2f ms" % (k, v))
print("sum %.2f" % sum(tot.values()))
