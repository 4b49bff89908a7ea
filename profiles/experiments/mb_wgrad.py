import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops
ops.set_precision("bf16")
which = sys.argv[2] if len(sys.argv) > 2 else "crn256"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
layers = {"crn256": (8, 1, 256, 256, 256), "vgg128": (16, 1, 128, 128, 128), "lstm1": (8, 24, 24, 48, 200), "enc": (32, 24, 24, 24, 100), "crn512_64": (8, 1, 512, 512, 64)}
N, G, Cin, Cout, S = layers[which]
x = torch.randn(N, G * Cin, S, S, device="cuda").requires_grad_(True)
w = (torch.randn(G * Cout, Cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
b = torch.zeros(G * Cout, device="cuda").requires_grad_(True)
prof = ops.KernelProfiler()
for i in range(reps + 1):
    if i == 1: ops.set_profiler(prof)
    y = ops.conv2d(x, w, b, stride=1, pad=1, act=1, slope=0.2, groups=G)
    y.backward(torch.ones_like(y))
ops.set_profiler(None)
for k, v in prof.summary().items():
    if v["flops"] > 0: print("%-40s %8.3f ms/launch %8.1f TF/s" % (k, v["ms"] / v["launches"], v["flops"] / v["ms"] / 1e9))
