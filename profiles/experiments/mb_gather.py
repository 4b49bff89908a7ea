"""North-star gather / blend kernels alone, at the benchmark batch (8) and at 8 GPUs' worth of samples (64): algorithmic GB/s."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops, synth
from oracle import torch_oracle as O
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
_, fidx = synth.body_mesh()
fi = torch.from_numpy(fidx.astype(np.int32)).cuda()
for B in (8, 64):
    S = 256
    b = synth.stage4_batch(77, min(B, 8))
    rep = lambda a: np.concatenate([a] * (B // a.shape[0]), 0) if a.shape[0] < B else a
    iuv = torch.from_numpy(rep(b["tgt_IUV255"])).cuda()
    tex = torch.randn(B, 72, 200, 200, device="cuda").requires_grad_(True)
    fg = float((iuv[..., 0] > 0).float().mean())
    # texture warp: IUV u8 + out + 4 taps x 3 ch x 4 B per foreground pixel
    bytes_tw = B * (S * S * 3 + 3 * S * S * 4 + fg * S * S * 48)
    out = ops.texture_warp(tex, iuv, False)
    g = torch.randn_like(out)
    t_f = timeit(lambda: ops.texture_warp(tex.detach(), iuv, False))
    def bwd():
        tex.grad = None
        o = ops.texture_warp(tex, iuv, False); o.backward(g)
    t_fb = timeit(bwd)
    # flow warp: src image + fim + wim + out
    verts = torch.from_numpy(rep(b["tgt_verts"])).cuda(); cam = torch.from_numpy(rep(b["tgt_cam"])).cuda()
    sv = torch.from_numpy(rep(b["src_verts"])).cuda()
    faces_t = ops.project_faces(verts, cam, fi, float(np.float32(O.EYE_Z)))
    faces_s = ops.project_faces(sv, cam, fi, float(np.float32(O.EYE_Z)))
    fim, wim = ops.rasterize_fim_wim(faces_t, S)
    img = torch.randn(B, 3, S, S, device="cuda")
    t_fw = timeit(lambda: ops.flow_warp(img, faces_s, fim, wim, None, False))
    bytes_fw = 4.0 * B * (3 * S * S + 4 * S * S + 3 * S * S)
    # blend
    r, bg, m = torch.randn(B, 3, S, S, device="cuda"), torch.randn(B, 3, S, S, device="cuda"), torch.rand(B, 1, S, S, device="cuda")
    t_bl = timeit(lambda: ops.blend(r, bg, m))
    bytes_bl = 4.0 * B * (3 + 3 + 1 + 3) * S * S
    t_ra = timeit(lambda: ops.rasterize_fim_wim(faces_t, S), 5)
    print("B=%2d  texture_warp fwd %6.1f us %6.0f GB/s | fwd+bwd %6.1f us | flow_warp %6.1f us %6.0f GB/s | blend %6.1f us %6.0f GB/s | rasterise %7.1f us"
          % (B, t_f * 1e3, bytes_tw / t_f / 1e6, t_fb * 1e3, t_fw * 1e3, bytes_fw / t_fw / 1e6, t_bl * 1e3, bytes_bl / t_bl / 1e6, t_ra * 1e3))
