"""Where a wave of conv_dma_kernel spends its cycles (probe build: conv_dma.hip compiled with -DCD_STAMP, see
conv_stamps_experiment.patch): shader-clock stamps around the barriers, the DMA issue, the wait for the DMA and the matrix-core steps,
summed over all waves of the launches."""
import sys, ctypes, torch
from jafpro_amd import ops
from jafpro_amd._lib import lib
ops.set_precision("bf16")
layers = {"crn256": (8, 1, [256], 256, 256), "crn64": (8, 1, [3, 512, 256], 512, 64), "vgg256_64": (16, 1, [256], 256, 64), "crn32": (8, 1, [3, 512, 256], 512, 32),
          "lstmlike": (8, 24, [36], 48, 200), "dec12": (8, 24, [12], 12, 200)}
names = ["barrier (top)", "DMA issue", "wait for DMA", "barrier (landed)", "matrix-core steps", "epilogue", "wave lifetime", "waves"]
L = lib()
L.jaf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, (N, G, cins, Cout, S) in layers.items():
    srcs = [torch.randn(N, G * c, S, S, device="cuda") for c in cins]
    w = torch.randn(G * Cout, sum(cins), 3, 3, device="cuda") * 0.1
    b = torch.randn(G * Cout, device="cuda")
    with torch.no_grad():
        f = lambda: ops.conv2d(srcs, w, b, stride=1, pad=1, act=1, slope=0.2, groups=G)
        f(); torch.cuda.synchronize()
        L.jaf_debug_stamps(None, 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        L.jaf_debug_stamps(out, 1)
    v = list(out)
    print("== %s  %.3f ms per launch, %d waves per launch, %.0f cycles per wave" % (name, e0.elapsed_time(e1) / 5, v[7] * 64 // 5, v[6] / max(v[7], 1)))
    for i in range(6):
        print("   %-20s %5.1f %%" % (names[i], 100.0 * v[i] / max(v[6], 1)))
    print("   %-20s %5.1f %%" % ("prologue / other", 100.0 * (v[6] - sum(v[:6])) / max(v[6], 1)))
