import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from jafpro_amd import ops
ops.set_precision("bf16")
N, G, Cin, Cout, S, K = 32, 24, 3, 12, 200, 5
x = torch.randn(N, G * Cin, S, S, device="cuda").requires_grad_(True)
w = (torch.randn(G * Cout, Cin, K, K, device="cuda") * 0.05).requires_grad_(True)
b = torch.zeros(G * Cout, device="cuda").requires_grad_(True)
prof = ops.KernelProfiler()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for i in range(reps + 1):
    if i == 1: ops.set_profiler(prof)
    y = ops.conv2d(x, w, b, stride=1, pad=K // 2, act=1, slope=0.2, groups=G)
    y.backward(torch.ones_like(y))
ops.set_profiler(None)
for k, v in prof.summary().items():
    print("%-48s %8.3f ms/launch" % (k, v["ms"] / v["launches"]))
