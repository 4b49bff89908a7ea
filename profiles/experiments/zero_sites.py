"""Counts zero-fill call sites (torch.zeros / zeros_like / Tensor.zero_) of one train step by caller line."""
import sys, os, torch, collections, traceback
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
cnt = collections.Counter()
def site():
    for fr in traceback.extract_stack()[::-1][2:]:
        if "jafpro_amd" in fr.filename or "bench.py" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"
oz, ozl, oz_ = torch.zeros, torch.zeros_like, torch.Tensor.zero_
def z(*a, **k):
    cnt["zeros " + site()] += 1; return oz(*a, **k)
def zl(*a, **k):
    cnt["zeros_like " + site()] += 1; return ozl(*a, **k)
def z_(self):
    cnt["zero_ " + site()] += 1; return oz_(self)
torch.zeros, torch.zeros_like, torch.Tensor.zero_ = z, zl, z_
tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
torch.zeros, torch.zeros_like, torch.Tensor.zero_ = oz, ozl, oz_
print("python-side zero fills per step:", sum(cnt.values()))
for k, v in cnt.most_common(40): print("%4d  %s" % (v, k))
