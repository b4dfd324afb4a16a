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


def refill(args, net, cfg):
    import numpy as np
    import torch
    from xiangqi_alphazero_amd import engine, evaluator
    from xiangqi_alphazero_amd.sample_format import RESULT_DTYPE
    ev, ev_name = evaluator.make_evaluator(net, "cuda", "hip" if args.evaluator == "auto" else args.evaluator)
    ecfg = engine.make_config(args.slots, args.sims, c_puct=cfg.c_puct, temperature_threshold=cfg.temperature_threshold,
                              max_game_length=cfg.max_game_length, random_opening_moves=cfg.random_opening_moves,
                              enable_resign=cfg.enable_resign, resign_threshold=cfg.resign_threshold,
                              resign_check_steps=cfg.resign_check_steps, add_noise=True, seed=11, games_target=1 << 30,
                              max_out_samples=args.slots * 201 * 2, max_out_results=args.slots * 8)
    eng = engine.SelfPlayEngine(ecfg, "cuda", evaluator=ev)
    eng.capture_step()
    t0 = time.time()
    marks, steps_all = [], []
    while True:
        for _ in range(256):
            eng.step()
        st = eng.stats()
        now = time.time() - t0
        _, res = eng.drain_device()                          # keep the rings empty; game lengths of the measured window
        if now >= args.warm_s and len(res):
            steps_all.append(res.cpu().numpy().reshape(-1).view(RESULT_DTYPE)["steps"].astype(np.int64))
        marks.append((now, int(st["games_finished"]), int(st["sims"])))
        if now >= args.warm_s + args.measure_s:
            break
    torch.cuda.synchronize()
    a = next(m for m in marks if m[0] >= args.warm_s)
    b = marks[-1]
    steps = np.concatenate(steps_all) if steps_all else np.zeros(1, dtype=np.int64)
    print(json.dumps({
        "config": {"mode": "refill (games_target unbounded: every finished slot starts a new game at once)", "slots": args.slots,
                   "sims_per_move": args.sims, "net": "%dx%d" % (args.channels, args.blocks), "evaluator": ev_name,
                   "launch": eng.launch_mode, "warm_s": args.warm_s, "measure_s": round(b[0] - a[0], 2)},
        "games_finished_in_window": b[1] - a[1], "games_per_hour": round((b[1] - a[1]) * 3600.0 / (b[0] - a[0]), 1),
        "simulations_per_s": round((b[2] - a[2]) / (b[0] - a[0]), 1),
        "plies_per_game_in_window": {"mean": round(float(steps.mean()), 2), "p10": int(np.percentile(steps, 10)),
                                     "p90": int(np.percentile(steps, 90))},
        "games_finished_before_window": a[1], "overflow": int(st["overflow"]), "samples_dropped": int(st["samples_dropped"])}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--evaluator", default="auto")
    ap.add_argument("--refill", action="store_true",
                    help="steady state of a REFILLING engine: finished slots start new games at once (games_target unbounded); the "
                         "rate is taken over --measure-s seconds after --warm-s seconds (several game lengths, so that the mix of "
                         "game ages is stationary: a fresh engine first finishes its SHORT games)")
    ap.add_argument("--warm-s", type=float, default=240.0)
    ap.add_argument("--measure-s", type=float, default=180.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    from xiangqi_alphazero_amd import model, selfplay, weights

    net = model.XiangqiNet(args.channels, args.blocks)
    net.load_state_dict(weights.make_state_dict(args.channels, args.blocks))
    cfg = Cfg()
    cfg.num_simulations = args.sims
    if args.refill:
        return refill(args, net, cfg)
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
