# Host enqueue time of one stage-4 train step vs its GPU time: run on the GPU box, output committed as
# profiles/round2_host_enqueue.txt.  "enqueue" = wall time of train_step() returning (everything queued), "total" = + synchronize.
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
import bench
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
for _ in range(2): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    tr.train_step(batch, next_batch=batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# the same step replayed from a captured hipGraph (Stage4Trainer.train_step_graphed)
for _ in range(2):          # a key is captured the second time it is seen in a row
    tr.train_step_graphed(batch, next_batch=batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    tr.train_step_graphed(batch, next_batch=batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("hipGraph replay: enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr.train_step(batch, next_batch=batch); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
