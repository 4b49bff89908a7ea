import sys, os, torch
sys.path.insert(0, "/root/repo")
from jafpro_amd import ops
ops.set_precision("bf16")
for (N, G, Cin, Cout, S) in ((32, 24, 3, 12, 200), (8, 24, 3, 12, 200)):
    x = torch.randn(N, G * Cin, S, S, device="cuda")
    w = (torch.randn(G * Cout, Cin, 5, 5, device="cuda") * 0.05).requires_grad_(True)
    b = torch.zeros(G * Cout, device="cuda").requires_grad_(True)
    prof = ops.KernelProfiler()
    for i in range(7):
        if i == 2: ops.set_profiler(prof)
        y = ops.conv2d(x, w, b, stride=1, pad=2, act=1, slope=0.2, groups=G)
        y.backward(torch.ones_like(y))
    ops.set_profiler(None)
    torch.cuda.synchronize()
    for k, v in prof.summary().items():
        if "wgrad" in k: print("N%d %-50s %8.3f ms/launch" % (N, k, v["ms"] / v["launches"]))
