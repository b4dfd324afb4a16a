"""Compact training-sample format of the engine and its adapter to the reference's dense schema.

The engine emits 640-byte `xq_sample` records (include/xq_hip.h): board, side to move, z, and the root's
(action, visit-count) pairs.  The reference's schema -- `(state float32[15,10,9], pi float64[8100], z float)`
doubled by the left-right mirror (training/parallel_selfplay.py:97-99, 123-151) -- is 70 KB per sample; it is
materialised here, on the consumer, with the same numpy operations the reference uses, so dense values are
bit-identical for T = 1 and agree to rounding of numpy's pow for T = 0.3.  This is data-format code (host side of
the boundary), not part of the search path.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

ACTION_SPACE = 8100

SAMPLE_DTYPE = np.dtype([("board", np.int8, 90), ("side", np.int8), ("z", np.int8), ("n_moves", np.uint8),
                         ("late_temp", np.uint8), ("ply", np.uint16), ("reserved0", np.uint16),
                         ("reserved1", np.uint16), ("slot", np.uint32), ("game_seq", np.uint32), ("pad", np.uint8, 20),
                         ("actions", np.uint16, 128), ("visits", np.uint16, 128)])
RESULT_DTYPE = np.dtype([("slot", np.uint32), ("game_seq", np.uint32), ("winner", np.int8), ("reason", np.uint8),
                         ("steps", np.uint16), ("n_samples", np.uint16), ("reserved", np.uint16)])
assert SAMPLE_DTYPE.itemsize == 640 and RESULT_DTYPE.itemsize == 16


def encode_planes(board: np.ndarray, side: int) -> np.ndarray:
    """XiangqiGame.get_state_for_nn (training/game.py:618-640) from a compact sample."""
    b = np.asarray(board, dtype=np.int8).reshape(10, 9)
    out = np.zeros((15, 10, 9), dtype=np.float32)
    for i in range(1, 8):
        out[i - 1] = (b == i * side)
        out[6 + i] = (b == -i * side)
    if side == 1:
        out[14] = 1.0
    return out


def dense_pi(actions: np.ndarray, visits: np.ndarray, temperature: float) -> np.ndarray:
    """MCTS._get_action_probs (training/mcts.py:190-206) from (action, visit) pairs, same numpy operations."""
    pi = np.zeros(ACTION_SPACE)
    pi[np.asarray(actions, dtype=np.int64)] = visits
    if temperature == 0:
        best = int(actions[int(np.argmax(visits))])
        pi = np.zeros(ACTION_SPACE)
        pi[best] = 1.0
    elif pi.sum() > 0:
        pi = pi ** (1.0 / temperature)
        pi /= pi.sum()
    return pi


def _flip_perm() -> np.ndarray:
    a = np.arange(ACTION_SPACE)
    frm, to = a // 90, a % 90
    fr, fc, tr, tc = frm // 9, frm % 9, to // 9, to % 9
    return ((fr * 9 + (8 - fc)) * 90 + (tr * 9 + (8 - tc))).astype(np.int64)


FLIP_PERM = _flip_perm()


def flip_sample(state: np.ndarray, pi: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """_augment_data (training/parallel_selfplay.py:137-151): mirror columns; pi'[flip(a)] = pi[a]."""
    flipped = np.zeros_like(pi)
    nz = np.nonzero(pi > 0)[0]
    flipped[FLIP_PERM[nz]] = pi[nz]
    return np.flip(state, axis=2).copy(), flipped


def to_reference_tuples(samples: np.ndarray, results: np.ndarray, late_temperature: float = 0.3,
                        augment: bool = True):
    """-> (all_data, per_game) where all_data is the reference's list of (state, pi, z) with each sample followed by
    its mirror image, games in result order, plies in order; per_game = [(winner, steps, n_samples)]."""
    all_data: List[tuple] = []
    per_game = []
    key = samples["slot"].astype(np.int64) << 32 | samples["game_seq"].astype(np.int64)
    for r in results:
        k = int(r["slot"]) << 32 | int(r["game_seq"])
        mine = samples[key == k]
        mine = mine[np.argsort(mine["ply"], kind="stable")]
        for s in mine:
            n = int(s["n_moves"])
            t = late_temperature if s["late_temp"] else 1.0
            state = encode_planes(s["board"], int(s["side"]))
            pi = dense_pi(s["actions"][:n], s["visits"][:n].astype(np.float64), t)
            z = float(s["z"])
            all_data.append((state, pi, z))
            if augment:
                fs, fp = flip_sample(state, pi)
                all_data.append((fs, fp, z))
        per_game.append((int(r["winner"]), int(r["steps"]), len(mine)))
    return all_data, per_game


def reachable_actions() -> np.ndarray:
    """Sorted action ids (from*90 + to) along which SOME piece on SOME square can ever move, whatever the position:
    rook/cannon/pawn/king lines (same row or column), knight jumps, one- and two-step diagonals (advisor, elephant;
    taken on the whole board, a superset of game.py:262-452 / game_core.pyx:185-330 for any placement, legal or not).
    2 550 of the 8 100 ids: the policy head only has to produce these columns for the engine, every other logit can
    never belong to a legal move."""
    acts = []
    for fr in range(10):
        for fc in range(9):
            for tr in range(10):
                for tc in range(9):
                    dr, dc = abs(tr - fr), abs(tc - fc)
                    if (dr, dc) == (0, 0):
                        continue
                    if dr == 0 or dc == 0 or (dr, dc) in ((1, 2), (2, 1), (1, 1), (2, 2)):
                        acts.append((fr * 9 + fc) * 90 + tr * 9 + tc)
    return np.array(sorted(acts), dtype=np.int64)
