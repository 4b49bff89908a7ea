# upper bound of what one grouped launch (or level-parallel streams) buys for the ConvLSTM cells of one time step:
# the five levels' cell launches one after the other on one stream vs each level on its own stream
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from jafpro_amd import ops
ops.set_precision("bf16")
N, G, T = 8, 24, 2
levels = [(12, 200), (24, 100), (24, 50), (48, 25), (96, 13)]
data = []
for C, S in levels:
    x = torch.randn(T, N, G * C, S, S, device="cuda")
    w = torch.randn(G * 4 * C, 2 * C, 3, 3, device="cuda") * 0.05
    b = torch.zeros(G * 4 * C, device="cuda")
    data.append((x, w, b))
streams = [torch.cuda.Stream() for _ in levels]
def run_serial():
    for x, w, b in data:
        ops.convlstm(x, w, b, groups=G, return_all=False)
def run_parallel():
    main = torch.cuda.current_stream()
    for (x, w, b), st in zip(data, streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            ops.convlstm(x, w, b, groups=G, return_all=False)
    for st in streams:
        main.wait_stream(st)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with torch.no_grad():
    for _ in range(2):
        print("T=%d steps, 5 levels: one stream %.3f ms   five streams %.3f ms" % (T, timeit(run_serial), timeit(run_parallel)))
    for (C, S), (x, w, b) in zip(levels, data):
        print("  level C%d @%d alone: %.3f ms" % (C, S, timeit(lambda: ops.convlstm(x, w, b, groups=G, return_all=False))))
