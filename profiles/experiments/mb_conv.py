import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
which = sys.argv[2] if len(sys.argv) > 2 else "all"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
def T(*s): return torch.randn(*s, device="cuda")
def timeit(name, fn, flops):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s %8.3f ms  %8.1f TF/s" % (name, ms, flops / ms / 1e9))
layers = {
 "crn256":  (8, 1, [256], 256, 256, 3, 1),
 "crn259":  (8, 1, [3, 128, 128], 256, 256, 3, 1),
 "crn512_64": (8, 1, [512], 512, 64, 3, 1),
 "vgg64":   (8, 1, [64], 64, 256, 3, 1),
 "dec4":    (8, 24, [24], 6, 200, 3, 1),
 "enc3":    (32, 24, [24], 24, 100, 3, 1),
 "inp36":   (8, 24, [36], 12, 200, 3, 1),
 "crn4":    (8, 1, [3, 512], 512, 4, 3, 1),
 "crn8":    (8, 1, [3, 512, 256], 512, 8, 3, 1),
 "crn8b":   (8, 1, [512], 512, 8, 3, 1),
 "crn16":   (8, 1, [3, 512, 256], 512, 16, 3, 1),
 "crn16b":  (8, 1, [512], 512, 16, 3, 1),
 "crn32":   (8, 1, [3, 512, 256], 512, 32, 3, 1),
 "crn32b":  (8, 1, [512], 512, 32, 3, 1),
 "crn64":   (8, 1, [3, 512, 256], 512, 64, 3, 1),
 "vgg512_16": (16, 1, [512], 512, 16, 3, 1),
 "vgg512_32": (16, 1, [512], 512, 32, 3, 1),
 "vgg256_64": (16, 1, [256], 256, 64, 3, 1),
 "unet8":   (8, 1, [256], 256, 8, 3, 1),
 "unet16":  (8, 1, [128], 128, 16, 3, 1),
 "lstm25":  (32, 24, [24], 24, 25, 3, 1),
 "lstm50":  (32, 24, [24], 24, 50, 3, 1),
 "dg48":    (8, 24, [48], 24, 200, 3, 1),
 "dg96":    (8, 24, [96], 48, 100, 3, 1),
 "enc1":    (32, 24, [3], 12, 200, 5, 1),
 "dec12":   (8, 24, [12], 12, 200, 3, 1),
 "dec72":   (8, 24, [72], 24, 100, 3, 1),
}
for name, (N, G, cins, Cout, S, k, st) in layers.items():
    if which != "all" and which != name: continue
    srcs = [T(N, G * c, S, S) for c in cins]
    w = T(G * Cout, sum(cins), k, k) * 0.1
    b = T(G * Cout)
    fl = 2.0 * N * G * Cout * sum(cins) * k * k * (S // st) ** 2
    with torch.no_grad():
        timeit(name + " fwd", lambda: ops.conv2d(srcs, w, b, stride=st, pad=k // 2, act=1, slope=0.2, groups=G), fl)
if which in ("all", "lstm1"):
    N, G, C, S, Tn = 8, 24, 12, 200, 2
    x = T(Tn, N, G * C, S, S); w = T(G * 4 * C, 2 * C, 3, 3) * 0.1; b = T(G * 4 * C)
    fl = 2.0 * N * G * 4 * C * (C + 2 * C) * 9 * S * S
    with torch.no_grad():
        timeit("lstm1 T=2 fwd", lambda: ops.convlstm(x, w, b, groups=G), fl)
