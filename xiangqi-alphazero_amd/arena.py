"""Arena gate -- the reference's model-vs-model evaluation (`AlphaZeroTrainer._serial_evaluate`,
training/train.py:453-535) on the device-resident engine: all `eval_games` games run concurrently (one slot
each), colours alternate by game index, every move is `MCTS.get_action(temperature=0, add_noise=False)` with
`eval_simulations`, a game still running after `max_game_length` plies is a draw, and
`win_rate = (new_wins + 0.5 draws) / games >= eval_win_rate` promotes the candidate.  Deterministic (no random
draws at all), so results are checked game by game against the reference's own arena under stub evaluators.

Per step every slot is expanded from ONE network's output -- the one whose side is searching (train.py:479-483).  With the
hand-written evaluators both models run over the whole slot batch (a few dozen games: the launches, not the FLOPs, are
the cost) with the other model's requests masked out, and the merge happens on the device: a step has no host round trip.
Dense-protocol evaluators (the stubs of the fixtures) are run on their own slots only.  With a process group the games are sharded over the
ranks like self-play games (`distributed.shard_games`); the per-game results are all-gathered and every rank derives the
same verdict from the same gathered table.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import os

import numpy as np
import torch

from . import engine, evaluator as ev_mod, hip
from .sample_format import RESULT_DTYPE


def _evaluate_subset(eng, ev, x, idx, policy_is_probs, dense, legal, value):
    """Run `ev` on the slots `idx` (int64 device tensor) and scatter its outputs into the full-width buffers."""
    if idx.numel() == 0:
        return
    xs = x.index_select(0, idx)
    if hasattr(ev, "evaluate_legal") and not policy_is_probs:
        ll, v = ev.evaluate_legal(xs, eng.req_moves.index_select(0, idx), eng.req_counts.index_select(0, idx))
        legal.index_copy_(0, idx, ll)
    else:
        p, v = ev(xs)
        dense.index_copy_(0, idx, p)
    value.index_copy_(0, idx, v.view(-1))


def play_arena(eval_new: Callable, eval_old: Callable, eval_games: int, eval_simulations: int, max_game_length: int,
               c_puct: float = 1.5, device="cuda", policy_is_probs: bool = False, first_game: int = 0):
    """eval_*: evaluators in either protocol (`evaluate_legal`, or a callable float32[n,15,10,9] -> (policy
    float32[n,8100], value float32[n])); both must use the same one.  Plays games first_game .. first_game+eval_games-1
    of the arena (the new model is red in even games) and returns the results array ordered by game (slot == game -
    first_game)."""
    cfg = engine.make_config(eval_games, eval_simulations, c_puct=c_puct, max_game_length=max_game_length,
                             random_opening_moves=0, enable_resign=False, add_noise=False, games_target=eval_games,
                             manual_moves=2)
    eng = engine.SelfPlayEngine(cfg, device)
    dev = eng.device
    new_is_red = ((torch.arange(eval_games, device=dev) + first_game) % 2 == 0)
    sparse = hasattr(eval_new, "evaluate_legal") and hasattr(eval_old, "evaluate_legal") and not policy_is_probs
    dense = None if sparse else torch.zeros((eval_games, 8100), dtype=torch.float32, device=dev)
    legal = torch.zeros((eval_games, 128), dtype=torch.float32, device=dev) if sparse else None
    value = torch.zeros(eval_games, dtype=torch.float32, device=dev)
    zero = torch.zeros_like(eng.req_counts)
    # XQ_ARENA_STREAMS=0: both networks on the one stream (A/B runs and the equality test)
    side = torch.cuda.Stream(device=dev) if sparse and os.environ.get("XQ_ARENA_STREAMS", "1") != "0" else None

    def step():
        x = eng.select()
        # the model that is SEARCHING evaluates every node of its search (root and leaves at any depth), so the
        # choice follows the side to move of the real game, not of the evaluated position (train.py:479-483)
        red_to_move = eng.slot_ints[:, 0] == 1
        use_new = new_is_red == red_to_move
        if sparse:
            # No host round trip in a step: each model runs over the whole (small) slot batch with the OTHER model's
            # request counts masked to zero -- xq_policy_head_legal skips those rows -- and the two results are merged
            # on the device.  Batch size and conv variant are the same every step (no nonzero(), no data-dependent shapes).
            # The two networks are independent: the old one runs on a side stream (forked from / joined to the main one), so at arena
            # batch sizes -- ten games, every kernel a fraction of the chip -- the two towers overlap; recorded into the graph as two
            # parallel branches.
            c_new, c_old = torch.where(use_new, eng.req_counts, zero), torch.where(use_new, zero, eng.req_counts)
            if side is not None:
                main = torch.cuda.current_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ll_old, v_old = eval_old.evaluate_legal(x, eng.req_moves, c_old)
                ll_new, v_new = eval_new.evaluate_legal(x, eng.req_moves, c_new)
                main.wait_stream(side)
            else:
                ll_new, v_new = eval_new.evaluate_legal(x, eng.req_moves, c_new)
                ll_old, v_old = eval_old.evaluate_legal(x, eng.req_moves, c_old)
            torch.where(use_new.unsqueeze(1), ll_new, ll_old, out=legal)
            torch.where(use_new, v_new.view(-1), v_old.view(-1), out=value)
            eng.expand_legal(legal, value)
        else:
            _evaluate_subset(eng, eval_new, x, use_new.nonzero().view(-1), policy_is_probs, dense, legal, value)
            _evaluate_subset(eng, eval_old, x, (~use_new).nonzero().view(-1), policy_is_probs, dense, legal, value)
            eng.expand(dense, value, policy_is_probs)

    # A sparse step is ~70 launches for a handful of games (eval_games = 10 in the reference's config): launch-bound.  It has no host
    # synchronisation and no data-dependent shape, so it is recorded once into a HIP graph and replayed (same kernels, same results);
    # XQ_ARENA_GRAPH=0 keeps it eager.  Only a capture-unsupported error keeps the eager path (as engine.capture_step).
    graph = None
    if sparse and os.environ.get("XQ_ARENA_GRAPH", "1") != "0":
        step()
        step()                                               # real steps: buffers of both evaluators exist before the recording
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                step()
            graph = g
        except hip.XqError:
            raise
        except RuntimeError as e:
            msg = str(e).lower()
            if not any(k in msg for k in ("captur", "hiperrorstreamcapture", "cudaerrorstreamcapture", "operation not permitted")):
                raise
            torch.cuda.synchronize(dev)
    while True:
        for _ in range(64):
            if graph is not None:
                graph.replay()
            else:
                step()
        st = eng.stats()
        if st["games_finished"] >= eval_games:
            break
    _, results = eng.drain()
    return results[results["slot"].argsort()]


def evaluate_models(new_model, old_model, config, device="cuda", evaluator_kind: str = "hip", group=None) -> Dict[str, object]:
    """Same stats dict as the reference (`new_wins, old_wins, draws, win_rate, model_updated`); reads
    `eval_games, eval_simulations, c_puct, max_game_length, eval_win_rate` from `config` (train.py:97-100).
    Under torch.distributed the games are split over the ranks and the winners all-gathered."""
    import torch.distributed as dist
    from . import distributed as xdist
    total = int(config.eval_games)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = xdist.shard_games(total, world, rank)
    first = sum(xdist.shard_games(total, world, r) for r in range(rank))
    winners = np.zeros(total, dtype=np.int64)
    steps = np.zeros(total, dtype=np.int64)
    if mine > 0:
        en, _ = ev_mod.make_evaluator(new_model, device, evaluator_kind)
        eo, _ = ev_mod.make_evaluator(old_model, device, evaluator_kind)
        res = play_arena(en, eo, mine, int(config.eval_simulations), int(config.max_game_length), float(config.c_puct),
                         device, first_game=first)
        winners[first:first + mine] = res["winner"].astype(np.int64)
        steps[first:first + mine] = res["steps"].astype(np.int64)
    if dist.is_initialized():         # disjoint shards: a sum gathers them; every rank ends with the same table (a group of
                                      # one rank runs the same collective)
        t = torch.from_numpy(np.stack([winners, steps])).to(device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, group=group)
        winners, steps = t[0].cpu().numpy(), t[1].cpu().numpy()
    new_wins = old_wins = draws = 0
    for game in range(total):
        w, new_is_red = int(winners[game]), game % 2 == 0
        if w == 0:
            draws += 1
        elif (w == 1) == new_is_red:
            new_wins += 1
        else:
            old_wins += 1
    win_rate = (new_wins + 0.5 * draws) / total
    games = np.zeros(total, dtype=RESULT_DTYPE)
    games["slot"], games["winner"], games["steps"] = np.arange(total), winners, steps
    return {"new_wins": new_wins, "old_wins": old_wins, "draws": draws, "win_rate": win_rate,
            "model_updated": win_rate >= float(config.eval_win_rate), "games": games}
