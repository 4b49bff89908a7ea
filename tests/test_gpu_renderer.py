"""the differentiable side of the renderer path on the GPU -- depth / alpha / face-inverse maps, the
neural_renderer silhouette and depth gradients (rasterize_cuda_kernel.cu:245-593) and the flow chain's adjoints
(projection, cal_bc_transform, grid_sample) -- against oracle/raster_oracle.c and torch autograd on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from jafpro_amd import ops
    return ops


def _mesh_faces(B, seed=91):
    from jafpro_amd import synth
    from oracle import torch_oracle as O
    v = synth.posed_vertices(seed, "v", B)
    cam = np.zeros((B, 3), np.float32); cam[:, 0] = 0.9
    _, fidx = synth.body_mesh()
    return v, cam, fidx, O.project_faces(torch.from_numpy(v), torch.from_numpy(cam), fidx).contiguous()


@pytest.mark.parametrize("S", [64, 256])
def test_rasterize_maps_bit_exact(S):
    from oracle import raster_oracle
    from jafpro_amd._lib import lib
    ops = _ops()
    _, _, _, faces = _mesh_faces(2)
    for flip in (0, 1):
        fim_r, wim_r, depth_r, finv_r = raster_oracle.rasterize_maps(faces.numpy(), S, 0.1, 100.0, flip=bool(flip))
        f = faces.cuda()
        B, NF = f.shape[0], f.shape[1]
        L = lib()
        ws = torch.empty(int(L.jaf_rasterize_workspace(B, NF, S)), device="cuda", dtype=torch.uint8)
        fim = torch.empty((B, S, S), device="cuda", dtype=torch.int32)
        wim = torch.empty((B, S, S, 3), device="cuda")
        depth = torch.empty((B, S, S), device="cuda")
        finv = torch.empty((B, S, S, 9), device="cuda")
        alpha = torch.empty((B, S, S), device="cuda")
        ops.check(L.jaf_rasterize_maps(ops._s(), ops._p(f), ops._p(fim), ops._p(wim), ops._p(depth), ops._p(finv), ops._p(alpha),
                                       ops._p(ws), B, NF, S, 0.1, 100.0, flip), "jaf_rasterize_maps")
        assert (fim.cpu().numpy() == fim_r).all()
        assert np.array_equal(wim.cpu().numpy(), wim_r) and np.array_equal(depth.cpu().numpy(), depth_r)
        assert np.array_equal(finv.cpu().numpy(), finv_r)
        assert np.array_equal(alpha.cpu().numpy(), (fim_r >= 0).astype(np.float32))
        assert (depth_r[fim_r < 0] == 100.0).all() and (fim_r >= 0).mean() > 0.1


def test_silhouette_gradient_known_answers_and_oracle():
    """The reference's two hand-checked silhouette gradients (tests/test_rasterize_silhouettes.py:37-99, rtol 1e-2)
    through the HIP Function, and grad_faces bit-exact vs the C restatement for a random upstream gradient on the body mesh."""
    from oracle import raster_oracle
    from oracle import torch_oracle as O
    ops = _ops()

    def kat(verts, pyi, pxi, minus1):
        v = torch.zeros(4, 3, 3)
        v[2] = torch.tensor(verts)
        v = v.cuda().requires_grad_(True)
        fi = torch.tensor([[0, 1, 2], [2, 1, 0]], device="cuda")
        eye = torch.tensor([0.0, 0.0, float(np.float32(O.EYE_Z))], device="cuda")
        faces = (v - eye)[:, fi].contiguous()                     # look_at from (0,0,eye_z): identity rotation
        img = ops.rasterize_silhouettes(faces, 64)
        torch.sum(torch.abs(img[:, pyi, pxi] - (1 if minus1 else 0))).backward()
        return v.grad[2].cpu().numpy()

    g1 = kat([[0.8, 0.8, 1.], [0.0, -0.5, 1.], [0.2, -0.4, 1.]], 25, 35, True)
    g2 = kat([[0.8, 0.8, 1.], [-0.5, -0.8, 1.], [0.8, -0.8, 1.]], 40, 50, False)
    r1 = np.array([[1.6725862, -0.26021874, 0.], [1.41986704, -1.64284933, 0.], [0., 0., 0.]], np.float32)
    r2 = np.array([[0.98646867, 1.04628897, 0.], [-1.03415668, -0.10403691, 0.], [3.00094461, -1.55173182, 0.]], np.float32)
    assert np.allclose(g1, r1, rtol=1e-2) and np.allclose(g2, r2, rtol=1e-2)

    _, _, _, faces = _mesh_faces(2)
    S = 128
    fim_r, wim_r, depth_r, finv_r = raster_oracle.rasterize_maps(faces.numpy(), S, flip=False)
    alpha_r = (fim_r >= 0).astype(np.float32)
    ga = np.random.default_rng(5).normal(size=alpha_r.shape).astype(np.float32)
    g_ref = raster_oracle.backward_pixel_map(faces.numpy(), fim_r, alpha_map=alpha_r, grad_alpha_map=ga, eps=1e-4)
    f = faces.cuda().requires_grad_(True)
    alpha, depth, fim, wim = ops.rasterize(f, S, 0.1, 100.0, 1e-4, True, False)
    (alpha * torch.from_numpy(ga).cuda()).sum().backward()
    assert float(np.abs(g_ref).max()) > 0
    assert np.array_equal(f.grad.cpu().numpy(), g_ref)


def test_depth_gradient_vs_oracle():
    from oracle import raster_oracle
    ops = _ops()
    _, _, _, faces = _mesh_faces(2)
    S = 128
    fim_r, wim_r, depth_r, finv_r = raster_oracle.rasterize_maps(faces.numpy(), S, flip=False)
    gd = np.random.default_rng(6).normal(size=depth_r.shape).astype(np.float32)
    g_ref = raster_oracle.backward_depth_map(faces.numpy(), depth_r, fim_r, finv_r, wim_r, gd, np.zeros_like(faces.numpy()))
    f = faces.cuda().requires_grad_(True)
    alpha, depth, fim, wim = ops.rasterize(f, S, 0.1, 100.0, 1e-4, False, True)
    (depth * torch.from_numpy(gd).cuda()).sum().backward()
    err = np.abs(f.grad.cpu().numpy() - g_ref).max() / np.abs(g_ref).max()
    assert err <= 1e-5, err                                      # fp32 atomics: summation order only


@pytest.mark.parametrize("border,ac", [(True, False), (True, True), (False, False)])
def test_grid_sample_backward(border, ac):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    src = (torch.rand(2, 3, 40, 37, generator=g) * 2 - 1).requires_grad_(True)
    grid = (torch.rand(2, 21, 19, 2, generator=g) * 2.6 - 1.3).requires_grad_(True)
    proj = torch.rand(2, 3, 21, 19, generator=g)
    ref = F.grid_sample(src, grid, mode="bilinear", padding_mode="border" if border else "zeros", align_corners=ac)
    (ref * proj).sum().backward()
    s2, g2 = src.detach().cuda().requires_grad_(True), grid.detach().cuda().requires_grad_(True)
    out = ops.grid_sample(s2, g2, border, ac)
    (out * proj.cuda()).sum().backward()
    assert (out.cpu() - ref).abs().max().item() <= 2e-6
    assert (s2.grad.cpu() - src.grad).abs().max().item() <= 1e-5
    assert (g2.grad.cpu() - grid.grad).abs().max().item() <= 1e-4 * max(1.0, grid.grad.abs().max().item())


def test_differentiable_flow_to_source_vertices():
    """d(warped image)/d(source SMPL vertices, camera) through projection -> cal_bc_transform -> grid_sample(border), the
    chain of src/cal_flow.py:28-39, vs torch autograd over the oracle's restatement with the same face-index / weight maps."""
    from oracle import raster_oracle
    from oracle import torch_oracle as O
    from jafpro_amd import synth
    ops = _ops()
    B, S = 2, 128
    vs, cam, fidx, _ = _mesh_faces(B, 92)
    vt, _, _, ft = _mesh_faces(B, 93)
    fim, wim = raster_oracle.rasterize_fim_wim(ft.numpy(), S)
    img = torch.from_numpy(synth.uniform(94, "img", (B, 3, S, S)))
    proj = torch.from_numpy(synth.uniform(94, "proj", (B, 3, S, S)))
    v_c = torch.from_numpy(vs).requires_grad_(True)
    c_c = torch.from_numpy(cam).requires_grad_(True)
    warped_ref, _ = O.flow_warp(img, O.project_faces(v_c, c_c, fidx), torch.from_numpy(fim), torch.from_numpy(wim))
    (warped_ref * proj).sum().backward()
    v_g = torch.from_numpy(vs).cuda().requires_grad_(True)
    c_g = torch.from_numpy(cam).cuda().requires_grad_(True)
    faces = ops.project_faces(v_g, c_g, torch.from_numpy(fidx).cuda(), float(np.float32(O.EYE_Z)))
    T = ops.bc_transform(faces, torch.from_numpy(fim).cuda(), torch.from_numpy(wim).cuda())
    warped = ops.grid_sample(img.cuda(), T, True, False)
    (warped * proj.cuda()).sum().backward()
    assert (warped.cpu() - warped_ref).abs().max().item() <= 1e-5
    gv, gc = v_c.grad, c_c.grad
    assert float(gv.abs().max()) > 1e-3
    assert (v_g.grad.cpu() - gv).abs().max().item() <= 1e-4 * float(gv.abs().max())
    assert (c_g.grad.cpu() - gc).abs().max().item() <= 1e-4 * float(gc.abs().max())


def test_smpl_renderer_silhouette_and_depth():
    """SMPLRenderer.render_silhouettes (src/nmr.py:295-310; anti-aliased as the reference constructs it) and render_depth
    vs the oracle composed the same way, forward and gradient to the vertices."""
    from oracle import raster_autograd as RA
    from oracle import torch_oracle as O
    from jafpro_amd import synth
    from jafpro_amd.nmr import SMPLRenderer
    v, cam, fidx, _ = _mesh_faces(1, 95)
    r = SMPLRenderer(faces=fidx, image_size=64, anti_aliasing=True).cuda()
    v_g = torch.from_numpy(v).cuda().requires_grad_(True)
    sil = r.render_silhouettes(torch.from_numpy(cam).cuda(), v_g)
    proj = torch.from_numpy(synth.uniform(95, "p", (1, 64, 64)))
    (sil * proj.cuda()).sum().backward()
    v_c = torch.from_numpy(v).requires_grad_(True)
    faces = O.project_faces(v_c, torch.from_numpy(cam), fidx)
    ref = F.avg_pool2d(RA.rasterize_silhouettes(faces, 128)[:, None], 2)[:, 0]
    (ref * proj).sum().backward()
    assert (sil.cpu() - ref).abs().max().item() == 0.0 and 0.1 < float(ref.mean()) < 0.9
    assert (v_g.grad.cpu() - v_c.grad).abs().max().item() <= 1e-5 * max(1.0, float(v_c.grad.abs().max()))
    r2 = SMPLRenderer(faces=fidx, image_size=64, anti_aliasing=False).cuda()
    d = r2.render_depth(torch.from_numpy(cam).cuda(), torch.from_numpy(v).cuda())
    dref = RA.rasterize_depth(O.project_faces(torch.from_numpy(v), torch.from_numpy(cam), fidx), 64)
    assert torch.equal(d.cpu(), dref)


def test_fused_flow_warp_is_bit_identical():
    """jaf_flow_warp_fwd == jaf_bc_transform -> jaf_grid_sample_fwd(border) -> jaf_mul_bcast, bit for bit, both align modes."""
    from oracle import raster_oracle
    from jafpro_amd import synth
    ops = _ops()
    B, S = 2, 128
    _, _, _, fs = _mesh_faces(B, 96)
    _, _, _, ft = _mesh_faces(B, 97)
    fim, wim = raster_oracle.rasterize_fim_wim(ft.numpy(), S)
    fim, wim, fs = torch.from_numpy(fim).cuda(), torch.from_numpy(wim).cuda(), fs.cuda()
    img = torch.from_numpy(synth.uniform(98, "img", (B, 3, 96, 80))).cuda()
    m3 = torch.from_numpy(synth.uniform(98, "m3", (B, 3, S, S), 0, 1)).cuda()
    for ac in (False, True):
        T = ops.bc_transform(fs, fim, wim)
        ref = ops.grid_sample(img, T, True, ac)
        assert torch.equal(ops.flow_warp(img, fs, fim, wim, None, ac), ref)
        assert torch.equal(ops.flow_warp(img, fs, fim, wim, m3, ac), ops.mul_bcast(ref, m3))
        assert torch.equal(ops.flow_warp(img, fs, fim, wim, m3[:, :1].contiguous(), ac), ops.mul_bcast(ref, m3[:, :1].contiguous()))
