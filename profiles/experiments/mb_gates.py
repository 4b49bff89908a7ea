import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops
from jafpro_amd._lib import lib
L = lib()
def p(t): return ctypes.c_void_p(t.data_ptr()) if t is not None else None
for (N, G, C, S) in ((8, 24, 12, 200), (8, 24, 24, 100), (8, 24, 48, 50)):
    HW = S * S
    dh = torch.randn(N, G * C, S, S, device="cuda"); dcn = torch.randn_like(dh); cp = torch.randn_like(dh); cc = torch.randn_like(dh)
    gates = torch.rand(N, G * 4 * C, S, S, device="cuda").to(torch.bfloat16)
    dcp = torch.empty_like(dh); packed = torch.empty(N * G * (4 * C // 8) * HW * 16, device="cuda", dtype=torch.uint8); db = torch.zeros(G * 4 * C, device="cuda")
    def run():
        rc = L.jaf_convlstm_gates_bwd_packed(ops._s(), N, G, C, HW, p(dh), p(dcn), p(gates), 1, p(cp), p(cc), p(dcp), p(packed), p(db))
        assert rc == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = N * G * C * HW * (4.0 * 5 + 8.0) + packed.numel()
    print("C=%d %dx%d: %.3f ms  %.0f GB/s" % (C, S, S, ms, nbytes / ms / 1e6))
