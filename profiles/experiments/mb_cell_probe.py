import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jafpro_amd import ops
ops.set_precision("bf16")
N, G, T = 8, 24, 4
for C, S in ((12, 200), (24, 100)):
    x = torch.randn(T, N, G * C, S, S, device="cuda")
    w = (torch.randn(G * 4 * C, 2 * C, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    b = torch.zeros(G * 4 * C, device="cuda", requires_grad=True)
    for dbg in (0, 512, 1024, 67, 64):
        os.environ["JAF_DBG"] = str(dbg)
        for rep in range(2):
            prof = ops.KernelProfiler(); ops.set_profiler(prof)
            h, _ = ops.convlstm(x, w, b, groups=G, return_all=False, return_state=False)
            ops.set_profiler(None)
            s = prof.summary()
        tot = sum(v["ms"] for k, v in s.items() if "true, false, false" in k)
        n = sum(v["launches"] for k, v in s.items() if "true, false, false" in k)
        print("C%d @%d dbg %2d: cell kernels %.3f ms / %d launches = %.1f us" % (C, S, dbg, tot, n, tot / n * 1e3))
        del h
