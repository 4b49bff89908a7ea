"""The stage-4 caller loop (counterpart of train/4.convLSTM_flowpro_interval.py:201-203, 249-261, 515-544).

The reference's script owns three things around the step that `step.Stage4Trainer.train_step` takes as inputs:

  * the loader loop (:201-203) -- here any iterable of batches, either the uint8 form the dataset decodes
    (`data.stage4_batch_from_uint8` turns it into the step's tensors on the device) or ready batch dicts;
  * the per-iteration draw of the reference subset and of the propagation source (:249-261): with probability 1/4 each,
    1, 2, 3 or 4 of the 4 reference frames in the order `np.random.choice(4, k, replace=False)` returns them, and the
    propagation source one of the drawn frames (`draw_subset` consumes a `numpy.random.RandomState` with exactly the
    calls of the script, so `RandomState(seed)` reproduces what `np.random.seed(seed)` gives the script);
  * the checkpoint cadence (:515-544): `count` starts at 12000, one file per module every `model_save_interval` counts,
    named as the script names them (`stages.save_checkpoints`), every module back to .train() afterwards.

The loop keeps one clip in flight ahead of the step: iteration k is given iteration k+1's batch and propagation source, so
the next clip's SMPL / renderer / frozen-network preparation runs beside this clip's loss backward (north star).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import data as jdata
from .stages import save_checkpoints
from .step import Stage4Trainer, _to_dev

START_COUNT = 12000             # train/4...py:197
SAVE_INTERVAL = 3000            # opt['model_save_interval'], :93
CKPT_MODULES = ("accu", "inpaint", "D", "face", "bg", "refine", "flow")      # the seven files of :518-533


def draw_subset(rng: np.random.RandomState, n_refs: int = 4) -> Tuple[Tuple[int, ...], int]:
    """One iteration's (used, prosrc), train/4...py:249-261, call for call: one `random()`, one `choice(4, k, replace=False)`
    and -- for k > 1 only -- one `choice(k, 1)`.  `used` keeps the drawn (unsorted) order: it is the time order the
    accumulate network's ConvLSTMs see (:269-276)."""
    r = rng.random_sample()
    k = 1 if r < 0.25 else 2 if r < 0.5 else 3 if r < 0.75 else 4
    idx = rng.choice(n_refs, k, replace=False)
    prosrc = int(idx[0]) if k == 1 else int(idx[rng.choice(k, 1)][0])
    return tuple(int(i) for i in idx), prosrc


def _as_batch(item: Dict, device, noise_gen: Optional[torch.Generator]) -> Dict:
    """A loader item as the step wants it: uint8 items go through the device input pipeline, host arrays are uploaded, and
    the background noise of :231 (`torch.randn(bg_mask.shape).cuda()`) is drawn here when the item does not carry it."""
    if "src_texture_u8" in item:
        raw = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(device) if isinstance(v, np.ndarray) and k not in ("face_bbox", "tgt_IUV_host") else v)
               for k, v in item.items()}
        b = jdata.stage4_batch_from_uint8(raw)
    else:
        b = _to_dev(item, device)                    # host arrays are uploaded, device tensors and host integers pass through
    if "bg_noise" not in b:
        s = b["src_img"]
        b["bg_noise"] = torch.randn((s.shape[0], 3, s.shape[-2], s.shape[-1]), device=s.device, dtype=torch.float32, generator=noise_gen)
    return b


def run_stage4(trainer: Stage4Trainer, loader: Iterable[Dict], iters: Optional[int] = None, ckpt_dir: Optional[str] = None,
               seed: Optional[int] = None, draws: Optional[Iterable[Tuple[Sequence[int], int]]] = None,
               save_interval: int = SAVE_INTERVAL, start_count: int = START_COUNT, epochs: int = 1,
               on_step: Optional[Callable[[int, Dict, Tuple[int, ...], int, Dict], None]] = None, graphed: bool = False,
               device="cuda") -> List[Dict]:
    """Runs the stage-4 loop on `trainer`: `epochs` passes over `loader` (the script: 2000, :193), at most `iters` iterations
    (epochs=None together with `iters`: the loader is cycled until they are done).

    seed / draws: the subset draws come from `numpy.random.RandomState(seed)` through `draw_subset` (seed=None: an unseeded
    state, like the script), or from `draws`, an explicit iterable of (used, prosrc) pairs.  With several ranks every rank must
    draw the same subsets -- the reference's DataParallel replicas share one draw -- so rank 0's seed is taken by all.
    on_step(count, losses, used, prosrc, batch): called after every iteration (logging, tests).
    ckpt_dir: where the seven `<prefix>_iter_<count>.pth` files go whenever count % save_interval == 0 (:515-533; rank 0 only).
    Returns one record per iteration: {"count", "used", "prosrc", losses as 1-element device tensors} -- reading a loss value
    synchronises the device, which the loop itself never does."""
    red = trainer.reducer
    multi = red is not None and red.active
    if draws is None:
        if multi:
            seed = red.host_broadcast_ints([np.random.randint(0, 2 ** 31 - 1) if seed is None else int(seed)])[0]
        rng = np.random.RandomState(seed)
        draw_it: Iterator = iter(lambda: draw_subset(rng), None)
    else:
        draw_it = iter(draws)
    rank = torch.distributed.get_rank() if multi else 0
    gen = None
    if seed is not None:
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed) + 7919 * rank)         # rank-local noise, as the replicas' shards are rank-local data

    def batches():
        e = 0
        while epochs is None or e < epochs:          # epochs=None: keep cycling the loader until `iters` is reached
            n = 0
            for item in loader:
                n += 1
                yield _as_batch(item, device, gen)
            e += 1
            if n == 0:
                return

    M = trainer.M
    mods = {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "D": M.discriminator, "face": M.F_Discriminator, "flow": M.propagater}
    step_fn = trainer.train_step_graphed if graphed else trainer.train_step
    history: List[Dict] = []
    count = start_count
    it = batches()
    cur = next(it, None)
    cur_draw = next(draw_it, None) if cur is not None else None
    done = 0
    while cur is not None and cur_draw is not None and (iters is None or done < iters):
        last = iters is not None and done + 1 >= iters
        nxt = None if last else next(it, None)
        nxt_draw = next(draw_it, None) if nxt is not None else None
        if nxt_draw is None:
            nxt = None
        count += 1                                                                   # :203
        used, prosrc = tuple(int(u) for u in cur_draw[0]), int(cur_draw[1])
        if prosrc not in used:
            raise ValueError("the propagation source %d is not among the used references %s (train/4...py:252-261)" % (prosrc, used))
        out = step_fn(cur, used=used, prosrc=prosrc, next_batch=nxt, next_prosrc=None if nxt is None else int(nxt_draw[1]))
        rec = {"count": count, "used": used, "prosrc": prosrc}
        rec.update({k: v for k, v in out.items() if k != "final_output"})
        history.append(rec)
        if on_step is not None:
            on_step(count, out, used, prosrc, cur)
        if ckpt_dir is not None and count > 0 and save_interval > 0 and count % save_interval == 0:    # :515
            if rank == 0:
                save_checkpoints(ckpt_dir, count, {k: mods[k] for k in CKPT_MODULES})
            for m in mods.values():                                                 # :536-542 (bg_model.train() included)
                m.train()
        cur, cur_draw = nxt, nxt_draw
        done += 1
    return history


def _synthetic_loader(n: int, B: int, seed: int = 1300, raw: bool = False):
    from . import synth
    for i in range(n):
        yield synth.stage4_raw(seed + i, B) if raw else synth.stage4_batch(seed + i, B)


def main(argv=None):
    """`python -m jafpro_amd.train --synthetic --iters 8`: the loop on synthetic clips (there is no dataset offline)."""
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--synthetic", action="store_true", help="synthetic clips (jafpro_amd.synth); the only data source offline")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", default="bf16", choices=["f32", "bf16", "bf16x3", "mixed"])
    ap.add_argument("--ckpt-dir", default=None)
    ap.add_argument("--save-interval", type=int, default=SAVE_INTERVAL)
    args = ap.parse_args(argv)
    if not args.synthetic:
        raise SystemExit("only --synthetic data is available offline (dataset I/O is the caller's: SURVEY section 6)")
    from . import ops, synth
    from .step import Stage4Models
    ops.set_precision(args.precision)
    _, fidx = synth.body_mesh()
    M = Stage4Models(fidx)
    for i, m in enumerate((M.Accu_model, M.inpaint_model, M.bg_model, M.refine_model, M.propagater, M.discriminator,
                           M.F_Discriminator, M.loss_criterion)):
        synth.load_synth(m, 1301 + i)
    tr = Stage4Trainer(M.cuda())
    hist = run_stage4(tr, _synthetic_loader(args.iters, args.batch), iters=args.iters, ckpt_dir=args.ckpt_dir, seed=args.seed,
                      save_interval=args.save_interval)
    torch.cuda.synchronize()
    for r in hist:
        print("count %d used %s prosrc %d total %.4f D %.4f G %.4f" % (r["count"], r["used"], r["prosrc"], float(r["total_loss"].reshape(-1)[0]),
                                                                      float(r["errD"].reshape(-1)[0]), float(r["errG"].reshape(-1)[0])))


if __name__ == "__main__":
    main()
