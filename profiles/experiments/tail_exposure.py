"""How much of the step does the LAST weight gradient of the backward pass cost?  The accumulate net's enc_0 (5 x 5, 3 -> 12 channels,
N = 32) is the first layer of the forward pass, so its weight gradient is the last kernel of the backward pass and nothing is left to
overlap it.  Timing probe only (the gradients of that layer are wrong while it is skipped): the step with and without that launch."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
orig = ops._conv_wgrad
skip = {"on": False, "n": 0}
def patched(ctx, m, weight, srcs, dz, dzp, inplace, stream=None):
    if skip["on"] and inplace and ((skip["k"] > 0 and m.KH == skip["k"]) or (skip["k"] == -1 and m.KH == 3 and m.G == 1 and m.H >= 128 and m.Cin >= 128)
                                   or (skip["k"] == -2 and m.KH == 3 and m.G == 24 and m.H >= 100)):
        skip["n"] += 1
        return None
    return orig(ctx, m, weight, srcs, dz, dzp, inplace, stream=stream)
ops._conv_wgrad = patched
def run(n=25):
    for _ in range(5): tr.train_step(batch, next_batch=batch)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): tr.train_step(batch, next_batch=batch)
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for rep in range(2):
    skip["on"] = False; a = run()
    skip["on"] = True; skip["k"] = 5; skip["n"] = 0; b = run(); n5 = skip["n"]
    skip["k"] = 7; skip["n"] = 0; c = run(); n7 = skip["n"]
    print("step %.2f ms; without the 5 x 5 weight gradients (%d launches skipped over 30 steps) %.2f ms; without the 7 x 7 ones (%d) %.2f ms" % (a, n5, b, n7, c))
    skip["k"] = -1; skip["n"] = 0; d = run(); nw = skip["n"]
    skip["k"] = -2; skip["n"] = 0; e = run(); np_ = skip["n"]
    print("   without the wide 3 x 3 ones at >= 128^2 (mid-backward; %d) %.2f ms; without the 24-part 3 x 3 ones at >= 100^2 (%d) %.2f ms" % (nw, d, np_, e))
