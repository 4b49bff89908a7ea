"""ORACLE -- TEST INFRASTRUCTURE ONLY (tests/ and oracle/make_step_golden.py).

Compact digests of a train step's results, so that the GPU tests compare against fixtures made ONCE by the CPU
oracle in the build container (oracle/make_step_golden.py -> tests/golden/step_*.npz) instead of re-running the
oracle on the GPU box (the GPU suite must fit the driver's 1200 s limit).

A module's gradient (28-36 M values) cannot be committed; what is kept per module is
  * `sq`  : the sum of squares of EVERY parameter tensor's gradient (float64, reference state_dict order),
  * `val` : the gradient at SAMPLES fixed pseudo-random positions of the flat gradient vector (same order),
so a test can check (a) every tensor's gradient norm and (b) the relative L2 distance of the module's gradient on an
unbiased sample: sqrt(sum_s (g_gpu - g_ref)^2 / sum_s g_ref^2).  Frames are kept whole (the north-star bar is an
L-infinity bar over every pixel), large intermediate tensors as strided samples.
"""
from __future__ import annotations

from typing import Dict, Iterable

import numpy as np

SAMPLES = 32768
TENSOR_SAMPLES = 16384


def sample_index(numel: int, salt: int = 0) -> np.ndarray:
    """Sorted positions (int64) into a flat vector of `numel` values; all of them when it is short."""
    if numel <= SAMPLES:
        return np.arange(numel, dtype=np.int64)
    rng = np.random.default_rng([int(numel), int(salt), 0x5EED])
    return np.sort(rng.choice(numel, SAMPLES, replace=False)).astype(np.int64)


def strided(a, n: int = TENSOR_SAMPLES) -> np.ndarray:
    """<= n values of a tensor at a fixed stride over its flattened contents (stride odd: no image-row aliasing)."""
    flat = np.asarray(a, dtype=np.float32).reshape(-1)
    return flat[::stride_for(flat.size, n)].copy()


def stride_for(numel: int, n: int = TENSOR_SAMPLES) -> int:
    s = max(1, numel // n)
    return s if s % 2 == 1 or s == 1 else s + 1


def flat_of(tensors: Iterable) -> np.ndarray:
    return np.concatenate([np.asarray(t, dtype=np.float32).reshape(-1) for t in tensors])


def digest(tensors: Dict[str, np.ndarray], idx: np.ndarray) -> Dict[str, np.ndarray]:
    """{'sq': per-tensor sum of squares, 'val': samples of the flat vector} for an ordered {key: array}."""
    sq = np.array([float((np.asarray(t, dtype=np.float64) ** 2).sum()) for t in tensors.values()], np.float64)
    return {"sq": sq, "val": flat_of(tensors.values())[idx]}
