"""B3 -- the self-play operator behind the reference's own call shape.

    all_data, stats = parallel_self_play(model, config, num_workers=None, use_gpu_server=False, gpu_device='cuda')

Same arguments, return schema and stats keys as training/parallel_selfplay.py:264-334, so
`AlphaZeroTrainer._parallel_self_play` (training/train.py:313-327) can call it unchanged.  What runs underneath is
the device-resident engine (`engine.SelfPlayEngine`): all `num_games_per_iter` games advance concurrently on one
GPU, the network is evaluated on whole leaf batches, and nothing is pickled or sent over a socket.
`num_workers` / `use_gpu_server` are accepted for signature compatibility and only recorded in the stats.
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import engine, evaluator
from .sample_format import to_reference_tuples

_CFG_KEYS = ("num_simulations", "c_puct", "temperature_threshold", "max_game_length", "random_opening_moves",
             "enable_resign", "resign_threshold", "resign_check_steps")     # parallel_selfplay.py:184-187


def run_games(model, config, num_games: int, device="cuda", n_slots: Optional[int] = None, seed: int = 0, rank: int = 0,
              evaluator_kind: str = "hip", poll_every: int = 64, device_records: bool = False, use_graph: bool = True):
    """Play `num_games` complete games; returns (samples, results, stats dict, elapsed seconds) in compact form:
    structured numpy arrays, or -- `device_records` -- uint8 device tensors [n, 640] / [m, 16] that never left the GPU."""
    slots = int(n_slots or min(num_games, 8192))
    slots = max(1, min(slots, num_games))
    ev, ev_name = evaluator.make_evaluator(model, device, evaluator_kind)
    cfg = engine.make_config(
        slots, int(config.num_simulations), c_puct=float(config.c_puct),
        temperature_threshold=int(config.temperature_threshold), max_game_length=int(config.max_game_length),
        random_opening_moves=int(config.random_opening_moves), enable_resign=bool(config.enable_resign),
        resign_threshold=float(config.resign_threshold), resign_check_steps=int(config.resign_check_steps),
        add_noise=True, seed=seed, rank=rank, games_target=num_games,
        max_out_samples=num_games * 201, max_out_results=num_games + 8)
    eng = engine.SelfPlayEngine(cfg, device, evaluator=ev)
    t0 = time.time()
    if use_graph and hasattr(ev, "evaluate_legal"):
        eng.capture_step()                             # one graph launch per step (short steps are launch-bound otherwise)
    while True:
        for _ in range(poll_every):
            eng.step()
        st = eng.stats()
        if st["games_finished"] >= num_games:
            break
    samples, results = eng.drain_device() if device_records else eng.drain()
    st = eng.stats()
    st["evaluator"] = ev_name
    st["launch"] = eng.launch_mode                     # "graph" (one HIP-graph replay per step) or "eager"
    if eng.capture_error:
        st["capture_error"] = eng.capture_error
    return samples, results, st, time.time() - t0


def parallel_self_play(model, config, num_workers: Optional[int] = None, use_gpu_server: bool = False,
                       gpu_device: str = "cuda", *, n_slots: Optional[int] = None, seed: int = 0,
                       return_compact: bool = False) -> Tuple[List[Tuple[np.ndarray, np.ndarray, float]], Dict[str, Any]]:
    for k in _CFG_KEYS + ("num_games_per_iter",):
        if not hasattr(config, k):
            raise AttributeError(f"config lacks '{k}' (see training/train.py:55-111)")
    num_games = int(config.num_games_per_iter)
    samples, results, st, elapsed = run_games(model, config, num_games, gpu_device, n_slots, seed)
    all_data, per_game = to_reference_tuples(samples, results, augment=True)
    wins = {1: 0, -1: 0, 0: 0}
    total_steps = 0
    for winner, steps, _n in per_game:
        wins[winner] += 1
        total_steps += steps
    stats = {
        "games": len(per_game), "red_wins": wins[1], "black_wins": wins[-1], "draws": wins[0],
        "avg_steps": total_steps / max(len(per_game), 1), "new_samples": len(all_data), "total_time": elapsed,
        "num_workers": int(st.get("games_started", num_games) and (n_slots or min(num_games, 8192))), "mode": "hip",
        "simulations": st["sims"], "leaf_evals": st["leaf_evals"], "root_evals": st["root_evals"],
        "evaluator": st["evaluator"], "launch": st["launch"],
    }
    if return_compact:
        stats["compact_samples"], stats["compact_results"] = samples, results
    return all_data, stats
