# which Python lines of the package issue the small ATen kernels of a train step (copies, fills, adds)?
import sys, os, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from torch.utils._python_dispatch import TorchDispatchMode
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
cnt = collections.Counter()
WATCH = ("copy_", "clone", "contiguous", "zeros", "zero_", "fill_", "add", "add_", "cat", "mul", "sum", "zeros_like", "full", "ones", "empty_like", "sub", "neg", "div", "_to_copy", "slice_backward", "select_backward", "new_zeros")
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=25)):
                if "jafpro_amd" in fr.filename and "aten_sites" not in fr.filename:
                    site = "%s:%d" % (os.path.basename(fr.filename), fr.lineno); break
            else:
                site = "autograd engine / torch"
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, s), c in cnt.items(): tot[n] += c
print("per op:", dict(tot))
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print("%4d  %-16s %s" % (c, n, s))
