# producers that write BOTH a packed bf16 image for their consumer AND an fp32 tensor: candidates for "bytes written for nobody"
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
def site():
    for fr in reversed(traceback.extract_stack(limit=30)):
        if "jafpro_amd" in fr.filename and os.path.basename(fr.filename) != "ops.py":
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"
orig_conv = ops.conv2d
def conv2d(srcs, weight, bias=None, *a, **k):
    y = orig_conv(srcs, weight, bias, *a, **k)
    if k.get("dst") is not None and k.get("keep_f32", True) and ops.packed_active():
        cnt[("conv", site(), tuple(y.shape))] += 1
    return y
ops.conv2d = conv2d
import jafpro_amd.networks as nw, jafpro_amd.crn_model as cm
orig_ln = ops.layernorm_lrelu
def ln(x, gamma, beta, eps=1e-5, slope=0.01, pre=None, dst=None, keep_f32=True, sole_consumer=False):
    if dst is not None and keep_f32 and ops.packed_active():
        cnt[("ln", site(), tuple(x.shape))] += 1
    return orig_ln(x, gamma, beta, eps, slope, pre, dst, keep_f32, sole_consumer)
ops.layernorm_lrelu = ln
orig_lstm = ops.convlstm
def lstm(x, weight, bias, groups=1, return_all=False, state=None, seq_image=None, final_dst=None, return_state=True):
    if seq_image is not None and return_all:
        cnt[("lstm all h fp32", site(), tuple(x.shape))] += 1
    return orig_lstm(x, weight, bias, groups, return_all, state, seq_image, final_dst, return_state)
ops.convlstm = lstm
tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
for (kind, s, shp), v in sorted(cnt.items(), key=lambda kv: -torch.Size(kv[0][2]).numel()):
    print("x%d %-16s %-22s %-28s %.1f MB fp32" % (v, kind, s, shp, torch.Size(shp).numel() * 4 / 1e6))
