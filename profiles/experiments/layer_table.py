import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
import bench
ops.set_precision("bf16")
_, fidx = synth.body_mesh()
M, mods = bench.build_models(fidx); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8), "cuda")
for _ in range(3): tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
last = [None]
orig = ops._make_desc
def md(N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, *a, **k):
    last[0] = (N, G, Cin, Cout, H, W, OH, OW, KH, stride, dil)
    return orig(N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, *a, **k)
ops._make_desc = md
class P(ops.KernelProfiler):
    def end(self, name, flops, ev0, nbytes=0.0):
        ev1 = torch.cuda.Event(enable_timing=True); ev1.record(torch.cuda.current_stream())
        self.records.append((name, float(flops), ev0, ev1, float(nbytes), last[0]))
p = P(); ops.set_profiler(p)
STEPS = 3
for it in range(STEPS):
    tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
ops.set_profiler(None)
agg = {}
for name, flops, e0, e1, nb, tag in p.records:
    r = agg.setdefault((name, tag), [0, 0.0, 0.0, 0.0])
    r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += flops; r[3] += nb
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values()) / STEPS
print("total timed %.2f ms/step" % tot)
for (name, tag), (n, ms, fl, nb) in rows[:int(os.environ.get('LT_ROWS', '70'))]:
    ms /= STEPS; n /= STEPS
    extra = ""
    if tag and fl:
        N, G, Cin, Cout, H, W, OH, OW, K, st, dil = tag
        inb = N * G * ((Cin + 7) // 8 * 8) * H * W * 2.0
        outb = N * G * Cout * OH * OW * 4.0
        extra = "N%d G%d %d->%d %dx%d k%d s%d d%d  %.0f TF  in+out %.0f MB -> %.0f GB/s(f32 out)" % (N, G, Cin, Cout, H, W, K, st, dil, fl / STEPS / n / (ms / n) / 1e9, (inb + outb) / 1e6, (inb + outb) / (ms / n) / 1e6)
    else:
        extra = "%.0f MB/launch %.0f GB/s  %s" % (nb / STEPS / n / 1e6, nb / STEPS / (ms) / 1e6, tag)
    print("%6.3f ms %4.0fx %-34s %s" % (ms, n, name[:34], extra))

# per kernel instantiation: launches, ms, algorithmic bytes (packed bf16 input once + fp32 output) per launch
by_name = {}
for (name, tag), (n, ms, fl, nb) in agg.items():
    if not (tag and fl): continue
    N, G, Cin, Cout, H, W, OH, OW, K, st, dil = tag
    inb = N * G * ((Cin + 7) // 8 * 8) * H * W * 2.0
    outb = N * G * Cout * OH * OW * 4.0
    r = by_name.setdefault(name, [0, 0.0, 0.0, 0.0])
    r[0] += n / STEPS; r[1] += ms / STEPS; r[2] += (inb + outb) * n / STEPS; r[3] += fl / STEPS
print("\nkernel instantiation: launches/step, ms/step, algorithmic MB/launch (bf16 in + fp32 out), GFLOP/launch")
for name, (n, ms, b, fl) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
    print("%-46s %5.0f %7.3f ms %8.1f MB %8.1f GF" % (name, n, ms, b / n / 1e6, fl / n / 1e9))
