"""Deterministic stub evaluator used to pin the MCTS against the reference.

`predict(state)` is a pure function of the 15x10x9 input planes and speaks the reference's evaluator
protocol (model.py:109-124): it returns (probs float32[8100], python float).  Only integer arithmetic and
single IEEE divisions are used, so every host (this container, the GPU box) produces bit-identical
arrays -- no exp/log whose SIMD implementation could differ between CPUs.

Two shapes:
  * flat   : near-uniform weights  (shallow, wide trees -- what a random-init net gives)
  * peaked : heavy-tailed weights  (deep trees, repeated visits, terminal leaves get reached)
"""
from __future__ import annotations

import zlib

import numpy as np

ACTION_SPACE = 8100
_IDX = np.arange(ACTION_SPACE, dtype=np.uint64)
_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * _M2
        x = (x ^ (x >> np.uint64(27))) * _M3
        return x ^ (x >> np.uint64(31))


def state_key(state: np.ndarray) -> int:
    planes = np.ascontiguousarray(state, dtype=np.float32)
    return zlib.crc32(planes.tobytes()) & 0xFFFFFFFF


def predict_from_key(key: int, peaked: bool):
    with np.errstate(over="ignore"):
        h = _mix((_IDX + np.uint64(1)) * _M1 + np.uint64(key))
    if peaked:
        r = (h >> np.uint64(40)) % np.uint64(64)          # 0..63
        w = np.uint64(1) + r * r * r * r                   # 1 .. ~1.6e7, heavy tail
    else:
        w = np.uint64(1024) + (h >> np.uint64(40)) % np.uint64(256)
    total = float(int(w.sum()))
    probs = (w.astype(np.float64) / total).astype(np.float32)
    v_int = int(_mix(np.array([key * 2654435761 + 12345], dtype=np.uint64))[0] % np.uint64(2001)) - 1000
    value = float(np.float32(v_int / 1000.0))              # what tensor.item() hands back for a float32
    return probs, value


class StubEvaluator:
    """Duck-typed stand-in for XiangqiNet / InferenceClient (`.predict(state[, device])`)."""

    def __init__(self, peaked: bool = False):
        self.peaked = peaked
        self.calls = 0

    def predict(self, state, device="cpu"):
        self.calls += 1
        return predict_from_key(state_key(state), self.peaked)

    def __call__(self, state):
        return self.predict(state)
