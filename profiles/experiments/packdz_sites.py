import sys, collections, torch
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
orig = ops._ConvFn._backward
def wrapped(ctx, dy):
    m = ctx.meta
    fused = getattr(ctx, "fused", None) is not None
    cnt[(fused, m.N, m.G, m.Cin, m.Cout, m.OH, m.OW, m.KH, m.stride, m.act, bool(ctx.needs_input_grad[0]), tuple(ctx.needs_input_grad[3:]))] += 1
    return orig(ctx, dy)
ops._ConvFn._backward = staticmethod(wrapped)
tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
tot = 0
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[0][1] * kv[0][2] * kv[0][4] * kv[0][5] * kv[0][6])):
    if not k[0]:
        tot += v
        print("x%2d  N%-3d G%-2d Cin%-4d Cout%-4d %3dx%-3d k%d s%d act%d wgrad=%s srcgrads=%s  dy %.1f MB" % (v, k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[11], k[1]*k[2]*k[4]*k[5]*k[6]*4/1e6))
print("unfused conv backward calls:", tot, " fused:", sum(v for k, v in cnt.items() if k[0]))
