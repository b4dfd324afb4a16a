"""Injected random draws shared by tests/golden/gen_golden.py (driving the reference) and the tests
(driving the oracle and the HIP engine).  Integer mixing + exact IEEE divisions only, so the same
draws come out on every host."""
from __future__ import annotations

import numpy as np

_MASK = (1 << 64) - 1


def _splitmix(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


class Stream:
    def __init__(self, seed: int, lane: int):
        self.key = _splitmix((seed << 8) ^ lane)
        self.ctr = 0

    def next_u64(self) -> int:
        self.ctr += 1
        return _splitmix(self.key ^ (self.ctr * 0xD1342543DE82EF95 & _MASK))

    def uniform(self) -> float:
        """in [0,1), multiple of 2^-53"""
        return (self.next_u64() >> 11) / float(1 << 53)


class Draws:
    """One independent stream per kind of draw, so the order in which the consumer interleaves the
    kinds does not matter."""

    def __init__(self, seed: int):
        self.s_randint = Stream(seed, 1)
        self.s_choice = Stream(seed, 2)
        self.s_dirichlet = Stream(seed, 3)
        self.s_uniform = Stream(seed, 4)

    def randint(self, lo: int, hi: int) -> int:            # random.randint (inclusive)
        return lo + self.s_randint.next_u64() % (hi - lo + 1)

    def choice_index(self, n: int) -> int:                  # random.choice(seq) -> index
        return self.s_choice.next_u64() % n

    def dirichlet(self, n: int) -> np.ndarray:              # stands in for np.random.dirichlet([0.3]*n)
        w = np.array([1 + (self.s_dirichlet.next_u64() >> 40) % 4096 for _ in range(n)], dtype=np.float64)
        w = w * w * w                                       # spiky, like alpha = 0.3
        return w / float(w.sum())

    def uniform(self) -> float:                             # the draw inside np.random.choice
        return self.s_uniform.uniform()
