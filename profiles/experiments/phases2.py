# main-stream time per phase: a step that starts on an EMPTY queue (right after a synchronize) vs a step enqueued behind another one
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
import bench
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
hp = ops.chain_stream(); hp.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(hp)
for _ in range(5): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
cold, warm, hostc = {}, {}, {}
R = 4
for rep in range(R):
    evs = []
    host = []
    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((name, e)); host.append((name, time.perf_counter()))
    tr.phase_mark = mark
    torch.cuda.synchronize()
    mark("start")
    tr.train_step(batch, next_batch=batch)          # cold: the queue is empty
    n_cold = len(evs)
    mark("start2")
    tr.train_step(batch, next_batch=batch)          # behind the first one
    mark("start3")
    tr.train_step(batch, next_batch=batch)
    torch.cuda.synchronize()
    for i in range(1, n_cold):
        cold[evs[i][0]] = cold.get(evs[i][0], 0.0) + evs[i - 1][1].elapsed_time(evs[i][1]) / R
        hostc[evs[i][0]] = hostc.get(evs[i][0], 0.0) + (host[i][1] - host[i - 1][1]) * 1e3 / R
    j0 = [i for i, (n, _) in enumerate(evs) if n == "start3"][0]
    for i in range(j0 + 1, len(evs)):
        warm[evs[i][0]] = warm.get(evs[i][0], 0.0) + evs[i - 1][1].elapsed_time(evs[i][1]) / R
print("%-30s %9s %9s %9s" % ("phase (ends at mark)", "cold GPU", "warm GPU", "host"))
for k in cold:
    print("%-30s %9.2f %9.2f %9.2f" % (k, cold[k], warm.get(k, float("nan")), hostc[k]))
print("%-30s %9.2f %9.2f %9.2f" % ("sum", sum(cold.values()), sum(warm.values()), sum(hostc.values())))
