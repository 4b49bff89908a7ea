import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jafpro_amd import ops
ops.set_precision("bf16")
N, G, T = 8, 24, 4
for C, S in ((12, 200), (24, 100), (24, 50)):
    x = torch.randn(T, N, G * C, S, S, device="cuda")
    w = (torch.randn(G * 4 * C, 2 * C, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    b = torch.zeros(G * 4 * C, device="cuda", requires_grad=True)
    for rep in range(3):
        prof = ops.KernelProfiler(); ops.set_profiler(prof)
        h, _ = ops.convlstm(x, w, b, groups=G, return_all=False, return_state=False)
        (h * h).sum().backward()
        ops.set_profiler(None)
        s = prof.summary()
    for k, v in sorted(s.items()):
        print("C%d @%d  %-52s %2d launches %8.1f us each" % (C, S, k[:52], v["launches"], v["ms"] / v["launches"] * 1e3))
    print("C%d @%d checksum %.6e" % (C, S, float(h.double().sum())))
