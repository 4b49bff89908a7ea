# where the time goes at the end of the backward pass / step boundary (events on the main stream)
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from jafpro_amd import ops, synth, step as S
from jafpro_amd.step import Stage4Trainer, _to_dev
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
b = _to_dev(synth.stage4_batch(1300, 8), "cuda")
hp = ops.chain_stream(); hp.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(hp)
evs = []
def mark(n):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((n, e))
# instrument: wrap FlatParams.adam and join
orig_adam = S.FlatParams.adam
def adam(self, lr, done=None):
    name = [k for k, v in tr.flat.items() if v is self][0]
    if done is None:
        ops.join_wgrad_stream(); mark("joined before Adam(%s)" % name)
    orig_adam(self, lr, done)
    mark("Adam(%s) issued" % name)
S.FlatParams.adam = adam
orig_fwd = M.Accu_model.forward_grouped
def fwd(x, T):
    mark("accu forward begins")
    y = orig_fwd(x, T)
    mark("accu forward done")
    return y
M.Accu_model.forward_grouped = fwd
tr.phase_mark = mark
if os.environ.get("NOJOIN"):
    _cnt = [0]
    orig_join = ops.join_wgrad_stream
    def jn():
        import traceback
        st = traceback.extract_stack(limit=3)
        if st[-2].name == "train_step":
            mark("(skipped final join)")
            return
        orig_join()
    ops.join_wgrad_stream = jn
orig_flush = S.flush_bn_counters
def fl(m):
    mark("before flush_bn_counters")
    orig_flush(m); mark("flush_bn_counters")
S.flush_bn_counters = fl
orig_zero = S.FlatParams.zero_grad
ZMODE = os.environ.get("Z", "zero")
def zg(self):
    if ZMODE == "zero": orig_zero(self)
    elif ZMODE == "fill": self.grad.fill_(0.0)
    elif ZMODE == "mul": self.grad.mul_(0.0)
    name = [k for k, v in tr.flat.items() if v is self][0]
    mark("zero_grad(%s)" % name)
S.FlatParams.zero_grad = zg
orig_a2p = ops.atlas_to_parts
def a2p(t):
    mark("before atlas_to_parts")
    y = orig_a2p(t); mark("atlas_to_parts"); return y
ops.atlas_to_parts = a2p
for _ in range(4): tr.train_step(b, next_batch=b)
torch.cuda.synchronize()
acc = {}
for rep in range(4):
    evs.clear()
    tr.train_step(b, next_batch=b); tr.train_step(b, next_batch=b); tr.train_step(b, next_batch=b)
    torch.cuda.synchronize()
    # take the SECOND step's tail and the third step's head
    idx = [i for i, (n, _) in enumerate(evs) if n == "D x3 updates"]
    lo = idx[1]
    seq = evs[lo:]
    for (n0, e0), (n1, e1) in zip(seq, seq[1:]):
        if n1 == "generator forward": break
        acc.setdefault(n1 if n1 not in acc or True else n1, []).append(e0.elapsed_time(e1))
        if n1 == "accu forward done": break
for k, v in acc.items():
    print("%-40s %7.3f ms" % (k, sum(v) / len(v)))
