# stand-alone timing of the fused up-sampling + packing pass of the CRN decoders (JAF_PACK_UP=0/1: general kernel / conv_pack_up_kernel)
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from jafpro_amd import ops
ops.set_precision("bf16")
for (N, cl, cp, cu, s) in ((8, 6, 64, 512, 64), (8, 6, 0, 256, 128), (8, 6, 64, 512, 64), (8, 6, 128, 512, 32), (8, 6, 128, 512, 16)):
    S = 2 * s
    label = torch.randn(N, cl, S, S, device="cuda")
    srcs = [label]
    if cp: srcs.append(torch.randn(N, cp, S, S, device="cuda"))
    low = torch.randn(N, cu, s, s, device="cuda")
    w = torch.randn(64, cl + cp + cu, 3, 3, device="cuda") * 0.02
    prof = ops.KernelProfiler()
    for i in range(7):
        if i == 2: ops.set_profiler(prof)
        up = ops.resize(low, (S, S), align_corners=True, lazy=True)
        y = ops.conv2d(srcs + [up], w, None, stride=1, pad=1, act=0)
    ops.set_profiler(None)
    torch.cuda.synchronize()
    for k, v in prof.summary().items():
        if "pack" in k: print("N%d [%d + %d + up %d] %d -> %d: %-40s %.3f ms/launch  (%.0f MB algorithmic, %.0f GB/s)" % (
            N, cl, cp, cu, s, S, k, v["ms"] / v["launches"], v["bytes"] / v["launches"] / 1e6, v["bytes"] / v["ms"] / 1e6))
