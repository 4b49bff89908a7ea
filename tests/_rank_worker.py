"""One data-parallel rank of the stage-4 trainer, started by tests/test_gpu_step_parity.py (one process per rank,
like the driver's torch.distributed.run launch).  JAF_RANK_BACKEND=gloo (default): the ranks talk over gloo and share
device 0 -- RCCL refuses two ranks on one device, and the arithmetic under test (per-rank BatchNorm statistics, gradient
means started from inside the backward pass, face-count weights) does not depend on the transport.  JAF_RANK_BACKEND=nccl:
RCCL, rank r on device r (test_two_rank_trainer_rccl, boxes with >= 2 GPUs).

  python tests/_rank_worker.py RANK WORLD PORT OUT.pt PRECISION SEED USED PROSRC [drop_face_rank]
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path, precision, seed = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
    used = tuple(int(c) for c in sys.argv[7].split(","))
    prosrc = int(sys.argv[8])
    drop_face_rank = int(sys.argv[9]) if len(sys.argv) > 9 else -1
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("JAF_RANK_BACKEND", "gloo")
    dev = rank if backend == "nccl" else 0            # RCCL: one device per rank; gloo: the ranks share device 0
    if backend == "nccl":
        from jafpro_amd.dist import limit_hw_queues
        limit_hw_queues()                             # before the first HIP call, as a multi-rank launcher should (dist.limit_hw_queues)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from jafpro_amd import ops, synth
    from jafpro_amd.dist import GradReducer, shard_batch
    from jafpro_amd.step import Stage4Trainer, _to_dev
    from tests._step_util import TRAINABLE, flat_in_reference_order, gpu_models, step_index
    M, mods = gpu_models()
    full = synth.stage4_batch(seed, world)
    if drop_face_rank >= 0:
        full["face_bbox"][drop_face_rank] = (96, 96, 32, 96)         # x0 == x1: no valid face on that rank
    shard = _to_dev(shard_batch(full, rank, world), "cuda")
    tr = Stage4Trainer(M, reducer=GradReducer(bucket_bytes=8 << 20))
    ops.set_precision(precision)
    out = tr.train_step(shard, used=used, prosrc=prosrc)
    torch.cuda.synchronize()
    # digests in the fixture's form (oracle/step_digest.py): samples of the flat gradient / parameter vectors at the
    # committed positions, per-tensor sums of squares, and the vectors' float64 sums (rank-equality checks)
    ix, digest = step_index(), {}
    for n in TRAINABLE:
        idx = torch.from_numpy(ix["idx." + n]).cuda()
        numel = torch.from_numpy(ix["numel." + n]).cuda()
        g, p = flat_in_reference_order(mods[n], n), flat_in_reference_order(mods[n], n, "data")
        ends = torch.cumsum(numel, 0)
        cs = torch.cat([torch.zeros(1, dtype=torch.float64, device="cuda"), torch.cumsum(g.double() ** 2, 0)])
        digest[n] = {"g": g[idx].cpu(), "p": p[idx].cpu(), "g_sq": (cs[ends] - cs[ends - numel]).cpu(),
                     "g_sum": float(g.double().sum()), "p_sum": float(p.double().sum())}
    res = {"losses": {k: float(v.reshape(-1)[0]) for k, v in out.items() if k != "final_output"},
           "final_output": out["final_output"].cpu(), "overlap_order": list(getattr(tr, "overlap_order", [])),
           "digest": digest, "backend": dist.get_backend(), "device": torch.cuda.current_device(),
           "buffers": {n: {k: v.cpu() for k, v in mods[n].state_dict().items() if "running_" in k or k.endswith("num_batches_tracked")}
                       for n in ("flow", "D", "face")}}
    torch.save(res, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
