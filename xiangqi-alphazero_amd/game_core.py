"""Drop-in for the reference's Cython plug point `game_core` (training/cython_engine/game_core.pyx:493-569).

Same five names, argument meaning, return types and error behaviour, so `training/game.py:36-42`
(`from game_core import cy_generate_legal_moves, ...`) binds to this module unchanged when it is first on
sys.path.  Every call runs the HIP kernels through the C ABI (include/xq_hip.h); a single board per call
is the degenerate batch -- real callers use the batched functions in `hip.py` / the engine.
No CPU fallback: without a GPU or without libxq_hip.so these functions raise.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip


def _board(board) -> torch.Tensor:
    # the reference's typed-buffer check: ndarray[signed char, ndim=2] -> ValueError otherwise (pyx:493)
    if not isinstance(board, np.ndarray):
        raise TypeError("Argument 'board' has incorrect type (expected numpy.ndarray)")
    if board.dtype != np.int8:
        raise ValueError("Buffer dtype mismatch, expected 'signed char'")
    if board.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % board.ndim)
    flat = np.ascontiguousarray(board).reshape(1, 90)   # copy: the caller's board is never mutated
    return torch.from_numpy(flat).cuda()


def _side(player: int, dev) -> torch.Tensor:
    return torch.tensor([int(player)], dtype=torch.int8, device=dev)


def cy_generate_legal_moves(board, player: int):
    """-> list of (from_row, from_col, to_row, to_col), in the reference's emission order (pyx:521-540)."""
    b = _board(board)
    moves, counts, _, _ = hip.movegen(b, _side(player, b.device))
    n = int(counts[0].item()) & 0xFFFF
    acts = moves[0, :n].cpu().numpy().astype(np.uint16)
    return [(int(a) // 90 // 9, int(a) // 90 % 9, int(a) % 90 // 9, int(a) % 90 % 9) for a in acts]


def cy_is_in_check(board, player: int) -> bool:
    """pyx:543-555 -- True when the king is missing."""
    b = _board(board)
    _, _, chk, _ = hip.movegen(b, _side(player, b.device))
    return bool(chk[0].item())


def cy_find_king(board, player: int):
    """pyx:493-505 -- (row, col) or None."""
    b = _board(board)
    k = int(hip.find_king(b)[0, 0 if player == 1 else 1].item())
    return None if k < 0 else (k // 9, k % 9)


def cy_is_attacked(board, kr: int, kc: int, by_player: int) -> bool:
    """pyx:508-518"""
    b = _board(board)
    return bool(hip.attack_map(b)[0, 0 if by_player == 1 else 1, kr * 9 + kc].item())


def cy_has_legal_moves(board, player: int) -> bool:
    """pyx:558-569"""
    b = _board(board)
    _, counts, _, _ = hip.movegen(b, _side(player, b.device))
    return (int(counts[0].item()) & 0xFFFF) > 0
