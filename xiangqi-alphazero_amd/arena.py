"""Arena gate -- the reference's model-vs-model evaluation (`AlphaZeroTrainer._serial_evaluate`,
training/train.py:453-535) on the device-resident engine: all `eval_games` games run concurrently (one slot
each), colours alternate by game index, every move is `MCTS.get_action(temperature=0, add_noise=False)` with
`eval_simulations`, a game still running after `max_game_length` plies is a draw, and
`win_rate = (new_wins + 0.5 draws) / games >= eval_win_rate` promotes the candidate.  Deterministic (no random
draws at all), so results are checked game by game against the reference's own arena under stub evaluators.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import engine, evaluator as ev_mod


def play_arena(eval_new: Callable, eval_old: Callable, eval_games: int, eval_simulations: int, max_game_length: int,
               c_puct: float = 1.5, device="cuda", policy_is_probs: bool = False):
    """eval_*: batched evaluators float32[G,15,10,9] -> (policy float32[G,8100], value float32[G]).
    Returns the per-game results array (slot == game index; new model plays red in even games)."""
    cfg = engine.make_config(eval_games, eval_simulations, c_puct=c_puct, max_game_length=max_game_length,
                             random_opening_moves=0, enable_resign=False, add_noise=False, games_target=eval_games,
                             manual_moves=2)
    eng = engine.SelfPlayEngine(cfg, device)
    new_is_red = (torch.arange(eval_games, device=eng.device) % 2 == 0)
    while True:
        for _ in range(64):
            x = eng.select()
            # the model that is SEARCHING evaluates every node of its search (root and leaves at any depth), so the
            # choice follows the side to move of the real game, not of the evaluated position (train.py:479-483)
            red_to_move = eng.slot_ints[:, 0] == 1
            use_new = new_is_red == red_to_move
            pn, vn = eval_new(x)
            po, vo = eval_old(x)
            policy = torch.where(use_new.unsqueeze(1), pn, po)
            value = torch.where(use_new, vn.view(-1), vo.view(-1))
            eng.expand(policy, value, policy_is_probs)
        st = eng.stats()
        if st["games_finished"] >= eval_games:
            break
    _, results = eng.drain()
    return results[results["slot"].argsort()]


def evaluate_models(new_model, old_model, config, device="cuda", evaluator_kind: str = "auto") -> Dict[str, object]:
    """Same stats dict as the reference (`new_wins, old_wins, draws, win_rate, model_updated`); reads
    `eval_games, eval_simulations, c_puct, max_game_length, eval_win_rate` from `config` (train.py:97-100)."""
    en, _ = ev_mod.make_evaluator(new_model, device, evaluator_kind)
    eo, _ = ev_mod.make_evaluator(old_model, device, evaluator_kind)
    res = play_arena(en, eo, int(config.eval_games), int(config.eval_simulations), int(config.max_game_length),
                     float(config.c_puct), device)
    new_wins = old_wins = draws = 0
    for r in res:
        w, new_is_red = int(r["winner"]), int(r["slot"]) % 2 == 0
        if w == 0:
            draws += 1
        elif (w == 1) == new_is_red:
            new_wins += 1
        else:
            old_wins += 1
    win_rate = (new_wins + 0.5 * draws) / int(config.eval_games)
    return {"new_wins": new_wins, "old_wins": old_wins, "draws": draws, "win_rate": win_rate,
            "model_updated": win_rate >= float(config.eval_win_rate), "games": res}
