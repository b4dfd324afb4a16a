#!/usr/bin/env python3
"""Measured (not derived) self-play games/hour: plays `--games` COMPLETE games with the engine on one GPU, under the
reference's termination rules, and reports games/hour, simulations/s and the game-length statistics that bench.py's
derived games/hour figure needs.  Default = BASELINE configs[1]: 1024 concurrent games, 400 sims/move, 128ch x 6blk.

    python tools/measure_games_per_hour.py [--games 1024 --slots 1024 --sims 400 --channels 128 --blocks 6]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Cfg:                                      # "full" preset of training/train.py:692-704, num_simulations from the CLI
    c_puct = 1.5
    temperature_threshold = 20
    max_game_length = 400
    random_opening_moves = 8
    enable_resign = True
    resign_threshold = -0.9
    resign_check_steps = 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--evaluator", default="auto")
    args = ap.parse_args()
    import numpy as np
    import torch
    from xiangqi_alphazero_amd import model, selfplay, weights

    net = model.XiangqiNet(args.channels, args.blocks)
    net.load_state_dict(weights.make_state_dict(args.channels, args.blocks))
    cfg = Cfg()
    cfg.num_simulations = args.sims
    t0 = time.time()
    samples, results, st, elapsed = selfplay.run_games(net, cfg, args.games, "cuda", n_slots=args.slots, seed=11,
                                                       evaluator_kind=args.evaluator, poll_every=256)
    torch.cuda.synchronize()
    steps = np.array([int(r["steps"]) for r in results])
    reasons = np.array([int(r["reason"]) for r in results])
    out = {
        "config": {"games": args.games, "slots": args.slots, "sims_per_move": args.sims,
                   "net": "%dx%d" % (args.channels, args.blocks), "evaluator": st["evaluator"]},
        "games_finished": int(len(results)), "wall_s": round(elapsed, 2),
        "games_per_hour": round(len(results) * 3600.0 / elapsed, 1),
        "simulations_per_s": round(st["sims"] / elapsed, 1),
        "plies_per_game": {"mean": round(float(steps.mean()), 2), "p10": int(np.percentile(steps, 10)),
                           "p50": int(np.percentile(steps, 50)), "p90": int(np.percentile(steps, 90)), "max": int(steps.max())},
        "endings": {"rules": int((reasons == 1).sum()), "max_length": int((reasons == 2).sum()), "resign": int((reasons == 3).sum())},
        "winners": {"red": st["red_wins"], "black": st["black_wins"], "draw": st["draws"]},
        "samples": int(len(samples)), "root_evals": st["root_evals"], "leaf_evals": st["leaf_evals"],
        "terminal_sims": st["terminal_sims"], "note": "games_target == games: the tail of the run has idle slots, so this "
        "under-states the steady-state rate of an engine that keeps refilling",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
