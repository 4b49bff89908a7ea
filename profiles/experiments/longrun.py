import sys, torch, time
sys.path.insert(0, ".")
import bench
from jafpro_amd import ops, synth
from jafpro_amd.step import Stage4Trainer, _to_dev
ops.set_precision(sys.argv[2] if len(sys.argv) > 2 else "bf16")
_, fidx = synth.body_mesh()
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 256
M, mods = bench.build_models(fidx, SIZE); M = M.cuda()
tr = Stage4Trainer(M)
batch = _to_dev(synth.stage4_batch(1300, 8, S=SIZE), "cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
for i in range(n + 1):
    out = tr.train_step(batch, next_batch=batch)
    if i % 100 == 0:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        print("step %4d allocated %.1f MB reserved %.1f MB peak_alloc %.1f MB segments %d inactive_split %.1f MB hipMalloc calls so far %d" % (
            i, torch.cuda.memory_allocated()/1e6, torch.cuda.memory_reserved()/1e6,
            torch.cuda.max_memory_allocated()/1e6, st["segment.all.current"], st["inactive_split_bytes.all.current"]/1e6,
            st["num_device_alloc"]), flush=True)
