"""Stage-4 train-step benchmark (BASELINE.json metric: train-step frames/sec, 256x256, stage 4).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one full stage-4 train step (generator forward, VGG+L1 loss, face-D update, 3 D updates,
generator backward, 6 Adam updates, gradient all-reduce when N > 1) over a synthetic batch of
B=8 samples per GPU that is already resident in HBM.  One target frame is generated per sample
(SURVEY F4), so frames/s = global batch / step time.  Default arithmetic is BASELINE configs[2]'s:
bf16 matrix-core operands, fp32 accumulation, bf16 storage of the tensors that only this library's kernels
read (--precision f32 / bf16x3 select the exact-fp32 and the split-bf16 parity-grade paths with fp32 tensors,
mixed a bf16x3 forward with a bf16 backward).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: dense MFMA peaks.  bf16x3 issues 3 bf16 MFMAs per algorithmic product.
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0 / 3.0, "mixed": 2500.0 / 3.0}
SEEDS = {"accu": 1301, "inpaint": 1302, "bg": 1303, "refine": 1304, "flow": 1305, "D": 1306, "face": 1307, "vgg": 1308}


def build_models(fidx, image_size=256, seeds=None):
    from jafpro_amd import synth
    from jafpro_amd.step import Stage4Models
    M = Stage4Models(fidx, image_size=image_size)
    mods = {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "flow": M.propagater, "D": M.discriminator, "face": M.F_Discriminator, "vgg": M.loss_criterion}
    for k, m in mods.items():
        synth.load_synth(m, (seeds or SEEDS)[k])
    return M, mods


# the weights / batch of the oracle's B=8 fixture tests/golden/step_s328_b8.npz (oracle/make_step_golden.py case s328_b8;
# tests/_step_util.SEEDS): the frame the fp32 CPU oracle generates for them is what `frame_linf_*` is measured against
FIXTURE_SEEDS = {"accu": 201, "inpaint": 202, "bg": 203, "refine": 204, "flow": 205, "D": 206, "face": 207, "vgg": 208}
FIXTURE_CASE, FIXTURE_BATCH_SEED, FIXTURE_B = "step_s328_b8.npz", 328, 8


def measure_frame_parity(fidx, modes):
    """L-inf distance of the generated frame from the fp32 CPU oracle's, per arithmetic mode, measured IN THIS RUN: the
    generator forward of the fixture's weights and B=8 batch on this GPU against the fixture's stored frame (north star: <= 1e-3
    for the parity-grade modes; bf16 rounds every operand to 8 bits and is held to 1e-1)."""
    from jafpro_amd import ops, synth
    from jafpro_amd.step import _to_dev, generator_forward
    path = os.path.join(ROOT, "tests", "golden", FIXTURE_CASE)
    if not os.path.exists(path):
        return None
    ref = torch.from_numpy(np.load(path)["final_output"]).cuda()
    M, _ = build_models(fidx, 256, FIXTURE_SEEDS)
    M = M.cuda()
    M.set_train_modes()
    b = _to_dev(synth.stage4_batch(FIXTURE_BATCH_SEED, FIXTURE_B), "cuda")
    out = {}
    prev = ops.get_precision()
    try:
        for mode in modes:
            ops.set_precision(mode)
            with torch.no_grad():
                g = generator_forward(M, b, (0, 1, 2, 3), 0)
            out[mode] = float((g["final_output"] - ref).abs().max())
    finally:
        ops.set_precision(prev)
    del M
    torch.cuda.empty_cache()
    return out


def cpu_baseline(mods, fidx, size=256):
    """The oracle (CPU restatement of the reference step, SURVEY 8(d) "CPU baseline beside it") timed on this node's
    host cores on a bounded sample: B=1 -- one warm-up + three timed full train steps, median -- and B=8 (the GPU
    workload's batch) -- one timed step.  At 512 x 512 (config 5 geometry, this build's own extension of the oracle): B=1,
    one warm-up + one timed step."""
    from jafpro_amd import synth
    from oracle.step_oracle import OracleStage4
    sds = {k: {kk: vv.detach().cpu().clone() for kk, vv in m.state_dict().items()} for k, m in mods.items()}
    # the box's CPU share, not the host's core count: oversubscribed OpenMP teams stall for minutes
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(share, 16)))
    orc = OracleStage4(sds, fidx)
    if size != 256:
        b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.stage4_batch(1400, 1, S=size).items()}
        orc.train_step(b)
        t0 = time.perf_counter()
        orc.train_step(b)
        dt = time.perf_counter() - t0
        return {"value": 1.0 / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "B=1: 1 full stage-4 train step after 1 warm-up (T=4, %dx%d fp32, %.1f s); reference-equivalent CPU path "
                          "(the oracle's size-parametrised form: the reference itself is 256-only), not optimised: torch CPU ops on "
                          "%d threads, brute-force C rasteriser single-threaded" % (size, size, dt, torch.get_num_threads())}
    b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.stage4_batch(1400, 1).items()}
    orc.train_step(b)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.train_step(b)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    b8 = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.stage4_batch(1300, 8).items()}
    t0 = time.perf_counter()
    orc.train_step(b8)
    dt8 = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "b8": {"value": 8.0 / dt8, "unit": "frames/s", "s_per_step": dt8},
            "sample": "B=1: median of 3 full stage-4 train steps after 1 warm-up (T=4, 256x256 fp32, %.1f s each); B=8 (the GPU "
                      "workload's batch): 1 step, %.1f s; reference-equivalent CPU path, not optimised: torch CPU ops on %d threads, "
                      "brute-force C rasteriser single-threaded" % (dt, dt8, torch.get_num_threads())}


def measure_ceilings():
    """Sustained dense bf16 MFMA rate and streaming-copy HBM rate of THIS GPU (include/jafpro_hip.h, ubench)."""
    import ctypes
    from jafpro_amd import ops
    from jafpro_amd._lib import lib
    L = lib()
    sink = torch.zeros(4, device="cuda")
    blocks, iters = 256 * 8, 4096
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.check(L.jaf_ubench_mfma_bf16(ops._s(), blocks, 64, ops._p(sink)), "jaf_ubench_mfma_bf16")
    e0.record()
    ops.check(L.jaf_ubench_mfma_bf16(ops._s(), blocks, iters, ops._p(sink)), "jaf_ubench_mfma_bf16")
    e1.record()
    torch.cuda.synchronize()
    tf = blocks * 4 * iters * 8 * 16384.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    n16 = (1 << 30) // 16
    src = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)
    dst = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)
    src.zero_()
    best, best_cfg = 0.0, None
    for variant, blocks in COPY_VARIANTS:
        ops.check(L.jaf_ubench_copy_variant(ops._s(), ops._p(src), ops._p(dst), n16, variant, blocks), "jaf_ubench_copy_variant")
        e0.record()
        for _ in range(5):
            ops.check(L.jaf_ubench_copy_variant(ops._s(), ops._p(src), ops._p(dst), n16, variant, blocks), "jaf_ubench_copy_variant")
        e1.record()
        torch.cuda.synchronize()
        gbps = 5 * 2.0 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        if gbps > best:
            best, best_cfg = gbps, (variant, blocks)
    return {"mfma_bf16_tflops": tf, "hbm_copy_GBps": best, "hbm_copy_variant": best_cfg,
            "how": "jaf_ubench_mfma_bf16: 2048 workgroups x 4 waves x 32768 register-fed v_mfma_f32_16x16x32_bf16 (the chip "
                   "lowers its clock under matrix-core load: MI355X_MICROARCH.md, DVFS); "
                   "jaf_ubench_copy_variant: 1 GiB read + 1 GiB written with 16-byte accesses, 5 passes, best of the "
                   "(loads in flight, nontemporal, grid) variants tried (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy)"}


# (variant, workgroups) of jaf_ubench_copy_variant tried by measure_ceilings: see include/jafpro_hip.h
COPY_VARIANTS = [(0, 4096), (1, 4096), (1, 16384), (2, 2048), (3, 16384), (4, 65536), (5, 262144)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults = what the driver passes (BENCH_r01.json): the caching allocator needs ~4 steps to reach its fixed point
    # (every step allocates ~40 GB of activations; until the block pattern repeats some requests fall through to hipMalloc)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--precision", default="bf16", choices=["f32", "bf16", "bf16x3", "mixed"],
                    help="matrix-core arithmetic of the convolutions: bf16 = BASELINE configs[2] (bf16 operands and bf16 storage of the "
                         "tensors between kernels, fp32 accumulate); f32 / bf16x3 = the parity-grade modes (fp32 tensors); mixed = "
                         "bf16x3 forward (frame and losses parity-grade), bf16 backward")
    ap.add_argument("--parity-mode-steps", type=int, default=3,
                    help="N=1 only: also time this many steps in the bf16x3 parity-grade mode (0 = skip)")
    ap.add_argument("--size", type=int, default=256, choices=[256, 512],
                    help="frame size: 256 = the BASELINE metric's configuration (configs[2]); 512 = configs[4] geometry "
                         "(no reference implementation at that size; throughput / roofline evidence, cpu_baseline = the "
                         "size-parametrised oracle at B=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--roofline-kernel", default=None,
                    help="report this kernel instantiation in `roofline` instead of the one with the largest total time")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the forward-only BASELINE configs[1] figure (N=1 only)")
    ap.add_argument("--no-frame-parity", action="store_true",
                    help="skip the in-run frame L-inf of each arithmetic mode against the oracle's B=8 fixture (N=1, 256 only)")
    ap.add_argument("--preheat", type=float, default=0.0,
                    help="seconds of synthetic matrix-multiply load before the warm-up steps (not steps; reported as preheat_s)")
    ap.add_argument("--no-prefetch", action="store_true", help="prepare each clip inside its own step instead of one step ahead")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: replay the step from a captured hipGraph (Stage4Trainer.train_step_graphed) instead of enqueuing "
                         "its ~1500 launches from Python every step")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the side-stream work on the main stream: per-kernel durations free of stream overlap "
                         "(the rocprofv3 summaries under profiles/ are taken with this flag; the roofline step always uses it)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if world > 1 or os.environ.get("JAF_BENCH_ONE_RANK_GROUP") == "1":
        from jafpro_amd.dist import limit_hw_queues
        limit_hw_queues()           # RCCL's own hardware queue on top of the step's five oversubscribes the GPU's (dist.limit_hw_queues)
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to stdout when its first communicator comes up, and any
    # library may chatter.  File descriptor 1 is pointed at stderr for the whole run; the result line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    if args.cpu_baseline_only:
        # child process of the default run (below): the CPU oracle alone, one JSON line
        from jafpro_amd import synth as _synth
        _, _fidx = _synth.body_mesh()
        _M, _mods = build_models(_fidx, args.size)
        emit(cpu_baseline(_mods, _fidx, args.size))
        return
    # CPU baseline first (rank 0, N=1), in a CHILD process started before this one touches the GPU: the oracle's 16 OpenMP
    # threads and its ~40 GB of host memory are gone when the timed loop starts (run in-process they left 80-95 ms
    # outliers in the first ten timed steps: profiles/round3_a_bench_bf16.json `step_ms`)
    cpu_result = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(args.size)],
                           capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise SystemExit("cpu baseline failed:\n" + r.stderr[-2000:])
        cpu_result = json.loads(lines[-1])
    # JAF_BENCH_BACKEND=gloo with JAF_BENCH_SHARE_GPU=1 lets several ranks share one GPU: the way the N>1 control
    # flow of this file is exercised on a 1-GPU box (RCCL refuses two ranks on one device); never a measurement.
    backend = os.environ.get("JAF_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("JAF_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    import torch.distributed as dist
    reducer = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        from jafpro_amd.dist import GradReducer
        reducer = GradReducer()
    elif os.environ.get("JAF_BENCH_ONE_RANK_GROUP") == "1":
        # experiment hook (profiles/experiments): the N>1 step -- gradient exchange started from inside backward, the joins of the
        # weight-gradient stream in front of every message, RCCL's own stream next to the step's side streams -- on the one GPU
        # of a gpurun box, with a process group of ONE rank.  The messages move nothing; what shows is the cost of the control flow.
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port_))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
        from jafpro_amd.dist import GradReducer
        reducer = GradReducer(skip_single=False)

    from jafpro_amd import ops, synth
    from jafpro_amd.step import Stage4Trainer, _to_dev
    ops.set_precision(args.precision)
    ops.set_serial_streams(bool(args.serial_streams))
    _, fidx = synth.body_mesh()
    M, mods = build_models(fidx, args.size)
    M = M.cuda()
    trainer = Stage4Trainer(M, reducer=reducer)
    B = args.batch
    batch = _to_dev(synth.stage4_batch(1300 + rank, B, S=args.size), "cuda")       # weak scaling: B per GPU fixed
    # algorithmic GFLOP per sample; at 512 the CRN / propagater / VGG terms scale by 4, the texture
    # networks stay at 200x200 and the image discriminator sees 256x256 (pooled)
    gflop_per_sample = 2278.8 if args.size == 256 else 6390.3

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # next_batch: the next clip's parameter-independent preparation (SMPL projection / rasteriser / flow
    # warp, frozen background CRN) is issued on the side HIP stream under this clip's loss backward.
    # Every step still performs exactly one preparation (the same synthetic clip is fed again).
    nb = None if args.no_prefetch else batch
    if not args.serial_streams:
        # the training loop runs under a HIGH-PRIORITY HIP stream (ops.chain_stream): the step's dependent chain then wins
        # the CUs whenever it competes with the side streams' work (weight gradients, next clip's preparation, perceptual
        # loss, weight re-packing).  Measured 63.63 -> 63.28 ms/step (two A/B pairs on one box).
        hp = ops.chain_stream()
        if hp is not None:
            hp.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(hp)
    step_fn = trainer.train_step
    if args.graph:
        if world > 1:
            raise SystemExit("--graph captures a single-rank step")
        step_fn = trainer.train_step_graphed
    if args.preheat > 0:
        # a freshly leased GPU: the first process to use it sees 80-95 ms steps scattered over its first second or two
        # (profiles/round3_a_bench_bf16_cpu_baseline_in_process.json, gpurun_out x1/x11 first runs; never in a second process
        # on the same box).  A plain matrix-multiply load before the warm-up steps, no part of the workload.
        _a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        _t = time.perf_counter()
        while time.perf_counter() - _t < args.preheat:
            for _ in range(20):
                _a @ _a
            torch.cuda.synchronize()
        del _a
    for _ in range(args.warmup):
        step_fn(batch, next_batch=nb)
    # the ~220 k Python objects built so far (modules, plans, caches) leave the garbage collector's scans: a full collection
    # costs the host 36 ms (profiles/experiments/gc_probe.py), as much as enqueueing a step, and the host is only a few
    # steps ahead of the GPU in the first timed steps
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host_ms = []
    for i in range(args.steps):
        _h = time.perf_counter()
        out = step_fn(batch, next_batch=nb)
        marks[i + 1].record()           # on the main stream, no synchronisation: per-step durations for the median
        host_ms.append((time.perf_counter() - _h) * 1e3)       # host time to ENQUEUE the step (no synchronisation inside)
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * B * args.steps / elapsed

    # host cost of one step: enqueue time with an EMPTY queue (inside the timed loop the host runs ahead of the GPU until the
    # HIP queues push back, so its per-step time there only mirrors the GPU's)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    step_fn(batch, next_batch=nb)
    host_enqueue_ms = (time.perf_counter() - h0) * 1e3
    barrier()

    result = {
        "metric": "train-step frames/sec, %dx%d 30-frame clips, stage-4" % (args.size, args.size),
        "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "median_ms_per_step": float(np.median(step_ms)), "preheat_s": args.preheat,
        "step_ms": [round(float(t), 2) for t in step_ms],
        "host_enqueue_step_ms": [round(float(t), 2) for t in host_ms],
        "frames_per_s_at_median": world * B / (float(np.median(step_ms)) * 1e-3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "stage-4 full train step (G fwd+bwd, VGG+L1, 3x D, face-D, 6x Adam), "
                               "B=%d/GPU, T=4 refs, %dx%d, 1 target frame/sample (BASELINE configs[%d]); "
                               "%s matrix-core arithmetic, fp32 accumulate; between convolutions packed bf16 images written by "
                               "the producing kernel; bf16 mode: ConvLSTM state / time-loop gradients, pre-LayerNorm tensors and "
                               "the gradients of image-only tensors stored in bf16, everything else fp32"
                               % (B, args.size, args.size, 2 if args.size == 256 else 4, args.precision),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": "dp%d" % world,
                   "clips_per_s": frames_per_s / 30.0,
                   "algorithmic_tflop_per_step": gflop_per_sample / 1e3 * world * B,
                   "launch_mode": "hipGraph replay" if args.graph else "eager, 4 HIP streams",
                   # Python + launch calls of one step on an empty queue (profiles/host_enqueue.py): the step is GPU-bound while
                   # this stays below ms_per_step
                   "host_enqueue_ms": host_enqueue_ms,
                   "loss": float(out["total_loss"].reshape(-1)[0])},
    }
    # what the gradient exchange ran on (scalars: proof for an N > 1 line that RCCL saw N ranks)
    rccl = {"world": world, "backend": dist.get_backend() if dist.is_initialized() else None,
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if (dist.is_initialized() and backend == "nccl") else None,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "bucket_mb": (reducer.bucket_elems * 4) >> 20 if reducer is not None else None,
            "overlap_order": list(getattr(trainer, "overlap_order", [])) or None}
    result["config"]["rccl"] = rccl
    result["config"].update({"rccl_world": rccl["world"], "rccl_backend": rccl["backend"], "rccl_version": rccl["rccl_version"],
                             "hw_queues": rccl["hw_queues"]})

    if rank == 0 and not args.no_roofline:
        # one extra step with every kernel on ONE stream: a kernel that shares the chip with side-stream work
        # runs longer than it does alone, and the roofline wants the kernel's own duration
        torch.cuda.synchronize()
        ops.set_serial_streams(True)
        trainer.reducer = None      # rank-0-only steps: no collective may be issued here (the other ranks wait below)
        trainer.train_step(batch, next_batch=nb)          # consumes the clip prepared on the side stream
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        trainer.train_step(batch, next_batch=nb)
        ops.set_profiler(None)
        torch.cuda.synchronize()
        ops.set_serial_streams(bool(args.serial_streams))
        allk = prof.summary()
        summ = {k: v for k, v in allk.items() if v["flops"] > 0}
        hbm = {k: v for k, v in allk.items() if v["bytes"] > 0}
        # dominant kernel = the template instantiation with the largest total time, named as rocprofv3 names it
        # (profiles/*kernel_stats*.csv carries the same rows)
        name, r = max(summ.items(), key=lambda kv: kv[1]["ms"])
        if args.roofline_kernel is not None:
            name, r = args.roofline_kernel, summ[args.roofline_kernel]
        achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
        tot_ms = sum(v["ms"] for v in summ.values())
        tot_fl = sum(v["flops"] for v in summ.values())
        peak = MFMA_PEAK_TFLOPS[args.precision]
        # HBM bytes per launch of the dominant kernel: not measurable inside this process (PMC counters need
        # their own rocprofv3 passes), so it is the committed figure of profiles/pmc_traffic.sh for that kernel
        # (pmc_traffic.json: 256 x 256; pmc_traffic_512.json: 512 x 512)
        traffic, traffic_src = None, None
        try:
            pj = "pmc_traffic.json" if args.size == 256 else "pmc_traffic_%d.json" % args.size
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pj)) as f:
                pt = json.load(f)
            if pt.get("kernel") == name and args.precision == "bf16" and pt.get("batch", 8) == B:
                traffic = pt["hbm_bytes_per_launch"]
                # where the committed figure comes from: the PMC table, the commit the library was at when it was taken, and
                # the git blob hashes of the kernel sources at that commit (a stale figure shows as a hash that no longer matches
                # `git hash-object` of the file: tests/test_host_logic.py checks it)
                traffic_src = {"file": pt["source"], "commit": pt.get("commit"), "kernel_source_blobs": pt.get("kernel_source_blobs")}
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                from pmc_summarize import git_blob_hash
                blobs = pt.get("kernel_source_blobs") or {}
                changed = sorted(p_ for p_, h_ in blobs.items() if git_blob_hash(os.path.join(ROOT, p_)) != h_)
                traffic_src["stale"] = (not blobs) or bool(changed)        # True: a listed source changed since the counters were read
                traffic_src["changed_since"] = changed
        except (OSError, ValueError, KeyError):
            pass
        ceil = measure_ceilings() if args.precision != "f32" else None
        result["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
            "kernel": name, "launches_per_step": r["launches"], "avg_launch_ms": r["ms"] / r["launches"],
            "algorithmic_gflop_per_launch": r["flops"] / r["launches"] / 1e9,
            # this box's own ceilings (register-fed MFMA loop, 16 B/lane copy) next to the nominal peaks
            "measured_ceilings": ceil,
            "frac_of_measured_mfma": (achieved / (ceil["mfma_bf16_tflops"] / (3.0 if args.precision == "bf16x3" else 1.0))) if ceil else None,
            "all_mfma_kernels": {"ms_per_step": tot_ms, "tflops": tot_fl / (tot_ms * 1e-3) / 1e12,
                                 "share_of_step": tot_ms / ms_per_step},
            "by_kernel": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in sorted(summ.items())},
            # the HBM-bound gather / blend / pack kernels of the same step: algorithmic bytes
            # / event time, against the 8 TB/s HBM3E peak
            "hbm_kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                "MB_per_launch": round(v["bytes"] / v["launches"] / 1e6, 2),
                                "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                "frac_of_8TBps": round(v["bytes"] / (v["ms"] * 1e-3) / 8e12, 3)} for k, v in sorted(hbm.items())},
        }
    if rank == 0 and world == 1 and args.parity_mode_steps > 0 and args.precision == "bf16":
        # the parity-grade mode (frame <= 1e-3 L-inf vs the fp32 oracle, tests/test_gpu_step.py) timed beside it
        # ... and the exact-fp32 arithmetic the north-star parity bar is stated in (v_mfma_f32_16x16x4_f32)
        for mode, key in (("mixed", "mixed_parity_mode"), ("bf16x3", "bf16x3_parity_mode"), ("f32", "f32_parity_mode")):
            ops.set_precision(mode)
            nsteps = args.parity_mode_steps if mode == "f32" else max(args.parity_mode_steps, 6)
            for _ in range(1 if mode == "f32" else 3):      # the allocator's block pattern changes with the mode: let it settle
                trainer.train_step(batch, next_batch=nb)
            torch.cuda.synchronize()
            # back to back, exactly like the main loop: the same clip fed again with its preparation issued one step ahead, no
            # synchronisation inside (per-step marks on the main stream for the median)
            mk = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
            t1 = time.perf_counter()
            mk[0].record()
            for i_ in range(nsteps):
                trainer.train_step(batch, next_batch=nb)
                mk[i_ + 1].record()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / nsteps
            ts = [mk[i_].elapsed_time(mk[i_ + 1]) * 1e-3 for i_ in range(nsteps)]
            trainer._prepared = None
            result["config"][key] = {"ms_per_step": dt * 1e3, "median_ms_per_step": float(np.median(ts)) * 1e3,
                                     "frames_per_s": B / dt, "steps": nsteps,
                                     "note": "back to back (no synchronisation between steps), next clip prepared one step ahead; "
                                             "mixed = bf16x3 forward (parity-grade frame and losses) + bf16 backward"}
        ops.set_precision(args.precision)
        # the same figures as scalar keys (a record that keeps only scalars of `config` still carries them)
        for mode in ("mixed", "bf16x3", "f32"):
            r_ = result["config"].get(mode + "_parity_mode")
            if r_:
                result["config"][mode + "_ms_per_step"] = r_["ms_per_step"]
                result["config"][mode + "_frames_per_s"] = r_["frames_per_s"]
    if rank == 0 and world == 1 and args.size == 256 and not args.no_frame_parity:
        fp = measure_frame_parity(fidx, ("f32", "bf16x3", "mixed", "bf16"))
        if fp is not None:
            for mode, v in fp.items():
                result["config"]["frame_linf_" + mode] = v
            result["config"]["frame_linf_fixture"] = "tests/golden/" + FIXTURE_CASE
    if rank == 0 and world == 1 and args.size == 256 and not args.no_config2:
        # BASELINE configs[1]: forward-only clip loop (test/conv_pro_test.py:219-279), B=2 clips x 30 target frames, fp32
        from jafpro_amd.step import forward_clip
        clip = _to_dev(synth.stage4_clip(1500, 2, 30), "cuda")
        figs = {}
        for mode in ("f32", "bf16x3", "bf16"):
            ops.set_precision(mode)
            with torch.no_grad():
                forward_clip(M, clip)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                forward_clip(M, clip)
                torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            figs[mode] = {"ms_per_2_clips": dt * 1e3, "frames_per_s": 60.0 / dt}
        ops.set_precision(args.precision)
        result["config"]["forward_only_config2"] = {
            "workload": "BASELINE configs[1]: forward-only, B=2 clips x 30 frames, 256x256 (9.90 algorithmic TFLOP per clip); "
                        "f32 and bf16x3 = the parity-grade arithmetics of tests/test_gpu_step.py::test_forward_clip_parity "
                        "(<= 1e-3 L-inf vs the fp32 oracle)", **figs}
    if cpu_result is not None:
        result["cpu_baseline"] = cpu_result
    if rank == 0:
        emit(result)
    if world > 1:
        dist.barrier()              # ranks > 0 wait here while rank 0 takes its roofline step
        dist.destroy_process_group()
    elif dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
