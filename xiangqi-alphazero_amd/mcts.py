"""Serving shim -- the reference's `MCTS` call shape (training/mcts.py:76-206) on the engine, for single-game callers
(the demos' `mcts.search(game, temperature=0.1, add_noise=False)`, `train.py`'s serial paths).

    mcts = MCTS(model, num_simulations=200, c_puct=1.5)
    pi = mcts.search(game, temperature=1.0, add_noise=True)      # float64[8100], same as the reference
    a  = mcts.get_action(game, temperature=0)

`game` is duck-typed like the reference's `XiangqiGame`: `.board` (int8[10,9]), `.current_player`, `.move_count`,
`.no_capture_count`, `.history` (list of the pre-move boards as 90-byte `bytes`).  `n_parallel` positions can be
searched at once with `search_many`.  One slot of a search-only engine (`manual_moves = 1`) per position; the network
is evaluated on the GPU.  With `add_noise=True` the root noise comes from the device Dirichlet(0.3) generator.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import engine, evaluator as ev_mod
from .sample_format import ACTION_SPACE, dense_pi


class MCTS:
    def __init__(self, model, num_simulations: int = 200, c_puct: float = 1.5, device: str = "cuda",
                 evaluator_kind: str = "auto", seed: int = 0):
        self.model = model
        self.num_simulations = num_simulations
        self.c_puct = c_puct
        self.device = device
        self.seed = seed
        self.evaluator = model if callable(model) and not hasattr(model, "state_dict") else \
            ev_mod.make_evaluator(model, device, evaluator_kind)[0]
        self._engines = {}

    def refresh(self, model=None):
        """Re-fold the weights after the caller changed its model in place (or hand over a new one).  The reference's
        MCTS reads the live model on every `.predict`; this shim snapshots BN-folded / pre-transformed weights, so a
        caller that keeps training between searches calls this first."""
        if model is not None:
            self.model = model
        if hasattr(self.evaluator, "update") and hasattr(self.model, "state_dict"):
            self.evaluator.update(self.model)

    def _engine(self, n: int, add_noise: bool):
        key = (n, add_noise)
        if key not in self._engines:
            cfg = engine.make_config(n, self.num_simulations, c_puct=self.c_puct, add_noise=add_noise, manual_moves=1,
                                     seed=self.seed)
            self._engines[key] = engine.SelfPlayEngine(cfg, self.device, evaluator=self.evaluator)
        return self._engines[key]

    def search_many(self, games: Sequence, temperature: float = 1.0, add_noise: bool = True) -> List[np.ndarray]:
        eng = self._engine(len(games), add_noise)
        for slot, g in enumerate(games):
            hist = [np.frombuffer(h, dtype=np.int8) for h in list(g.history)[-12:]]
            eng.set_position(slot, np.asarray(g.board, dtype=np.int8), int(g.current_player), int(g.move_count),
                             int(g.no_capture_count), np.stack(hist) if hist else None)
        for _ in range(self.num_simulations + 1):
            eng.step()
        out = []
        for slot in range(len(games)):
            r = eng.read_root(slot)
            if len(r["actions"]) == 0:
                out.append(np.zeros(ACTION_SPACE))                       # mcts.py:111-112
            else:
                out.append(dense_pi(r["actions"], r["visits"].astype(np.float64), temperature))
        return out

    def search(self, game, temperature: float = 1.0, add_noise: bool = True) -> np.ndarray:
        return self.search_many([game], temperature, add_noise)[0]

    def get_action(self, game, temperature: float = 0.0, add_noise: bool = False) -> int:
        """mcts.py:166-174"""
        probs = self.search(game, temperature, add_noise)
        if temperature == 0:
            return int(np.argmax(probs))
        return int(np.random.choice(len(probs), p=probs))
