"""Readers for the fixtures under tests/golden/ (written by tests/golden/gen_golden.py from the reference)."""
from __future__ import annotations

import json
import os
from functools import lru_cache

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@lru_cache(maxsize=None)
def corpus():
    z = np.load(os.path.join(GOLDEN, "corpus.npz"))
    return {k: z[k] for k in z.files}


@lru_cache(maxsize=None)
def crafted():
    z = np.load(os.path.join(GOLDEN, "crafted.npz"))
    return {k: z[k] for k in z.files}


def moves_of(d, i) -> np.ndarray:
    return d["moves_flat"][d["moves_off"][i]:d["moves_off"][i + 1]]


def attacked_bits(d, i) -> np.ndarray:
    """-> uint8[2][90]: attacked by red / by black"""
    return np.unpackbits(d["attacked"][i], axis=1)[:, :90]


@lru_cache(maxsize=None)
def perft():
    return {int(k): int(v) for k, v in json.load(open(os.path.join(GOLDEN, "perft.json"))).items()}


@lru_cache(maxsize=None)
def mcts_traces():
    return json.load(open(os.path.join(GOLDEN, "mcts_traces.json")))


@lru_cache(maxsize=None)
def game_traces():
    return json.load(open(os.path.join(GOLDEN, "game_traces.json")))


@lru_cache(maxsize=None)
def arena_traces():
    return json.load(open(os.path.join(GOLDEN, "arena_traces.json")))


def flip_perm() -> np.ndarray:
    return np.load(os.path.join(GOLDEN, "flip_perm.npy"))


def nn_golden():
    z = np.load(os.path.join(GOLDEN, "nn_golden.npz"))
    return {k: z[k] for k in z.files}


def nn_golden2():
    z = np.load(os.path.join(GOLDEN, "nn_golden2.npz"))
    return {k: z[k] for k in z.files}


def history_tail(d, i, n=12) -> np.ndarray:
    """Last n pre-move boards of the game position i belongs to (oldest first), int8[k,90], k<=n."""
    g, ply = d["game"][i], d["ply"][i]
    first = i - ply
    assert d["game"][first] == g and d["ply"][first] == 0
    lo = max(0, ply - n)
    return d["board"][first + lo:first + ply]


def hexf(x: str) -> float:
    return float.fromhex(x)
