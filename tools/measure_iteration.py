#!/usr/bin/env python3
"""One GPU, whole outer loop (SURVEY.md section 8 rows f1-f3 around the hot path; the reference's AlphaZeroTrainer.train,
training/train.py:581-637): two iterations of `train_loop.AlphaZeroLoop` -- self-play on the engine, the train step on the hand-written
kernels, the arena gate on iteration 2, checkpoints -- with the reference's default TrainingConfig (train.py:56-111: 128x6, batch 256,
5 epochs, buffer 50 000, 10 arena games at 100 simulations) except for the two sizes an 8-CPU box cannot reach and a GPU can:
`--games` concurrent self-play games per iteration (reference default 20) and `--sims` simulations per move (reference default 200).
Prints one JSON object with the wall time of every stage of every iteration.

    python tools/measure_iteration.py [--games 1024 --sims 100 --max-plies 120]
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--max-plies", type=int, default=300, help="max_game_length (reference default 300)")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--iterations", type=int, default=2)
    a = ap.parse_args()
    import torch
    from xiangqi_alphazero_amd import train_loop
    tmp = tempfile.mkdtemp(prefix="xq_iter_")
    cfg = types.SimpleNamespace(
        num_channels=a.channels, num_res_blocks=a.blocks, num_simulations=a.sims, c_puct=1.5, temperature_threshold=20,
        num_games_per_iter=a.games, max_game_length=a.max_plies, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True,
        random_opening_moves=4, num_iterations=a.iterations, batch_size=256, num_epochs=5, learning_rate=0.002, weight_decay=1e-4,
        lr_milestones=[50, 80], lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=500, eval_games=10, eval_win_rate=0.55,
        eval_simulations=100, checkpoint_dir=tmp, save_interval=1)
    loop = train_loop.AlphaZeroLoop(cfg, "cuda", seed=1)
    out = {"config": {k: v for k, v in vars(cfg).items() if k != "checkpoint_dir"},
           "native_train_step": bool(loop.current_model.res_blocks[0].native_conv), "iterations": []}
    for it in range(1, a.iterations + 1):
        loop.iteration = it
        t0 = time.perf_counter()
        sp = loop.self_play()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr = loop.train_network()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ev = loop.evaluate() if it % 2 == 0 and len(loop.buffer) >= cfg.min_buffer_size else {}
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        loop._save(it)
        t4 = time.perf_counter()
        n_batches = cfg.num_epochs * -(-len(loop.buffer) // cfg.batch_size)
        out["iterations"].append({
            "iteration": it, "self_play_s": round(t1 - t0, 2), "train_s": round(t2 - t1, 2), "arena_s": round(t3 - t2, 2),
            "checkpoint_s": round(t4 - t3, 2), "games": sp.get("games"), "samples_added": sp.get("new_samples"),
            "avg_plies": sp.get("avg_steps"), "buffer": len(loop.buffer), "games_per_hour_this_iteration": round(3600.0 * a.games / (t1 - t0), 1),
            "train_batches": n_batches, "train_samples_per_s": round(cfg.num_epochs * len(loop.buffer) / max(t2 - t1, 1e-9), 1),
            "policy_loss": tr.get("policy_loss"), "value_loss": tr.get("value_loss"), "evaluation": ev,
            "red_black_draw": [sp.get("red_wins"), sp.get("black_wins"), sp.get("draws")]})
    print(json.dumps(out, default=str))


if __name__ == "__main__":
    main()
