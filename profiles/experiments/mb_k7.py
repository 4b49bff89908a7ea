import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops
ops.set_precision("bf16")
for (Cin, Cout) in ((32, 1), (9, 32), (32, 3)):
    x = torch.randn(8, Cin, 262, 262, device="cuda").requires_grad_(True)
    w = (torch.randn(Cout, Cin, 7, 7, device="cuda") * 0.05).requires_grad_(True)
    b = torch.zeros(Cout, device="cuda").requires_grad_(True)
    prof = ops.KernelProfiler()
    for i in range(6):
        if i == 1: ops.set_profiler(prof)
        y = ops.conv2d(x, w, b, stride=1, pad=0, act=0, slope=0.0, groups=1)
        y.backward(torch.ones_like(y))
    ops.set_profiler(None)
    print("== %d -> %d k7" % (Cin, Cout))
    for k, v in prof.summary().items():
        print("   %-44s %3d x %8.3f ms/launch %8.1f TF/s" % (k, v["launches"], v["ms"] / v["launches"], v["flops"] / max(v["ms"], 1e-9) / 1e9))
