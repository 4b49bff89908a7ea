import sys, os, torch
os.environ["JAF_S2D_MIN_BLOCKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops
ops.set_precision("bf16")
torch.manual_seed(0)
for (N, G, Cin, Cout, H, W) in ((2, 3, 24, 24, 50, 50), (1, 2, 8, 16, 25, 25), (2, 24, 12, 24, 100, 100), (1, 1, 128, 72, 33, 31), (2, 1, 64, 32, 64, 64)):
    x = torch.randn(N, G * Cin, H, W, device="cuda")
    w = torch.randn(G * Cout, Cin, 3, 3, device="cuda") * 0.1
    res = []
    for s2d in (True, False):
        ops._S2D = s2d
        xd = x.clone().requires_grad_(True)
        y = ops.conv2d(xd, w, None, stride=2, pad=1, act=0, groups=G)
        gy = torch.ones_like(y) * 0.5 + torch.arange(y.numel(), device="cuda").reshape(y.shape).float().remainder(7.0) * 0.1
        y.backward(gy)
        res.append(xd.grad.clone())
    a, b = res
    d = (a - b).abs()
    print((N, G, Cin, Cout, H, W), "max diff %.3e of %.3e" % (d.max().item(), b.abs().max().item()), "frac > 1e-4:", (d > 1e-4 * b.abs().max()).float().mean().item())
    nz = (d > 1e-4 * b.abs().max()).nonzero()
    if nz.numel():
        print("   ", nz[:10].tolist())
        print("    y hist", torch.bincount(nz[:, 2] % 16, minlength=16).tolist(), "x hist", torch.bincount(nz[:, 3] % 16, minlength=16).tolist())
