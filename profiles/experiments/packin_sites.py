# which convolutions still pack their input from fp32 in a pass of their own (ops.pack_input), per train step
import sys, collections, traceback, os, torch
sys.path.insert(0, ".")
import bench
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
for _ in range(2): tr.train_step(batch, next_batch=batch)
cnt = collections.Counter()
orig = ops.pack_input
def wrapped(srcs, d, lazy=None):
    site = "?"
    for fr in reversed(traceback.extract_stack(limit=30)):
        if "jafpro_amd" in fr.filename and os.path.basename(fr.filename) != "ops.py":
            site = "%s:%d" % (os.path.basename(fr.filename), fr.lineno); break
    lz = lazy is not None and any(l is not None for l in (lazy if isinstance(lazy, (list, tuple)) else [lazy]))
    cnt[(site, d.N, d.G, d.Cin, d.H, d.W, len(srcs), lz)] += 1
    return orig(srcs, d, lazy)
ops.pack_input = wrapped
tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
tot = 0
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4] * kv[0][5] * kv[1]):
    site, N, G, Cin, H, W, ns, lz = k
    mb = N * G * Cin * H * W * 4 / 1e6
    tot += v
    print("x%2d %-22s N%-3d G%-2d Cin%-4d %3dx%-3d srcs %d lazy %d  fp32 in %.1f MB" % (v, site, N, G, Cin, H, W, ns, lz, mb))
print("pack_input calls per step:", tot)
