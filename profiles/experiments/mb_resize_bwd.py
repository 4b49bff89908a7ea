import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops
for (N, C, s, S) in ((8, 256, 128, 256), (8, 512, 64, 128), (8, 512, 32, 64), (8, 512, 16, 32), (8, 512, 8, 16), (8, 576, 100, 200), (8, 1152, 50, 100), (8, 2304, 25, 50)):
    x = torch.randn(N, C, s, s, device="cuda", requires_grad=True)
    y = ops.resize(x, (S, S), align_corners=True)
    g = torch.randn_like(y)
    def run():
        x.grad = None
        y.backward(g, retain_graph=True)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = 4.0 * N * C * (S * S + s * s)
    print("N%d C%d %d->%d: %.3f ms  %.0f GB/s" % (N, C, s, S, ms, nbytes / ms / 1e6))
