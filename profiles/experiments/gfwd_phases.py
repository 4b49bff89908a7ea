# main-stream time of the stages of generator_forward: right after a synchronize ("cold") and enqueued right behind a full train step ("warm")
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
for _ in range(5): tr.train_step(b, next_batch=b)
torch.cuda.synchronize()
used, prosrc = [0, 1, 2, 3], 0

def fwd(marks):
    def mark(n):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((n, e))
    with torch.no_grad():
        mark("start")
        prepared = S.prepare_clip(M, b, prosrc, True); mark("prepare issued (side stream)")
        x = ops.atlas_to_parts(b["src_texture_im"].contiguous()); mark("atlas_to_parts")
        accu = M.Accu_model.forward_grouped(x, 4); mark("accumulate net (24 parts, ConvLSTM)")
        masked = ops.part_mask_mul(accu, b["src_mask_im"].contiguous(), S._used_flags(4, used, accu.device)); mark("part mask")
        inpaint = M.inpaint_model.forward_grouped(masked); mark("inpaint net (24 parts)")
        iw = ops.texture_warp(inpaint, b["tgt_IUV255"], False); mark("texture warp")
        ro, fg = M.refine_model(iw, M.image_size); mark("refine CRN")
        torch.cuda.current_stream().wait_event(prepared.event); mark("wait prepared")
        fusion = ops.blend(ro, prepared.bg_output, fg)
        pro = M.propagater({"fake_tgt": fusion, "tsf_image": prepared.tsf, "use_mask": True, "tgt_smpl_mask": b["smpl_real_mask"], "tgt_IUV": b["tgt_IUV"], "use_IUV": True}); mark("blend + propagater")

res = {}
for mode in ("cold", "warm", "cold", "warm"):
    torch.cuda.synchronize()
    if mode == "warm":
        tr.train_step(b, next_batch=b)
    marks = []
    fwd(marks)
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        res.setdefault((mode, n1), []).append(e0.elapsed_time(e1))
names = [n for (m, n) in res if m == "cold"]
print("%-42s %8s %8s" % ("stage (no_grad forward)", "cold", "warm"))
for n in names:
    print("%-42s %8.2f %8.2f" % (n, min(res[("cold", n)]), min(res[("warm", n)])))
print("%-42s %8.2f %8.2f" % ("sum", sum(min(res[("cold", n)]) for n in names), sum(min(res[("warm", n)]) for n in names)))
