"""The outer training loop on top of the self-play path -- `AlphaZeroTrainer.train()` (training/train.py:581-638) with
the engine underneath, one process per GPU (BASELINE configs[4]).

Per iteration, exactly the reference's order:
  1. self-play     every rank plays `shard_games(num_games_per_iter, world, rank)` games on its own engine with the
                   current best weights (no collective during search), then `all_gather_samples` brings every rank's
                   compact records to every rank;
  2. train         rank 0 runs `train_network` on its device-resident replay buffer; the new weights reach the other
                   ranks as ONE flat `broadcast_weights`;
  3. arena gate    every second iteration (train.py:609) the candidate plays the best model; promote at
                   win_rate >= eval_win_rate, else the candidate reverts to the best weights (train.py:525-533);
  4. checkpoint    every `save_interval` iterations, the reference's file format; `training_stats.json` like train.py:620-634.
Works unchanged with world_size 1 (no process group needed).
"""
from __future__ import annotations

import copy
import json
import os
import time
from typing import Optional

import torch
import torch.distributed as dist

from . import arena, distributed as xdist, selfplay, training
from .model import XiangqiNet


class AlphaZeroLoop:
    def __init__(self, config, device="cuda", seed: int = 0, evaluator_kind: str = "auto"):
        self.config = config
        self.device = torch.device(device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.seed = seed
        self.evaluator_kind = evaluator_kind
        torch.manual_seed(seed)                        # identical initial weights on every rank
        self.current_model = XiangqiNet(config.num_channels, config.num_res_blocks).to(self.device)
        self.best_model = copy.deepcopy(self.current_model)
        self.optimizer = torch.optim.Adam(self.current_model.parameters(), lr=config.learning_rate,
                                          weight_decay=config.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=config.lr_milestones,
                                                              gamma=config.lr_gamma)
        self.buffer = training.ReplayBuffer(config.max_buffer_size, self.device)
        self.iteration = 0
        self.total_games = 0
        self.training_stats = []

    # ---- the three stages --------------------------------------------------------------------------------
    def self_play(self) -> dict:
        cfg = self.config
        mine = xdist.shard_games(cfg.num_games_per_iter, self.world, self.rank)
        t0 = time.time()
        if mine > 0:
            samples, results, st, _ = selfplay.run_games(self.best_model, cfg, mine, self.device,
                                                          seed=self.seed + 1000 * self.iteration, rank=self.rank,
                                                          evaluator_kind=self.evaluator_kind)
        else:
            import numpy as np
            from .sample_format import RESULT_DTYPE, SAMPLE_DTYPE
            samples, results = np.zeros(0, SAMPLE_DTYPE), np.zeros(0, RESULT_DTYPE)
        if self.world > 1:
            samples, results = xdist.all_gather_samples(samples, results, device=self.device)
        self.buffer.extend(samples)
        self.total_games += len(results)
        wins = {1: 0, -1: 0, 0: 0}
        for r in results:
            wins[int(r["winner"])] += 1
        return {"games": len(results), "red_wins": wins[1], "black_wins": wins[-1], "draws": wins[0],
                "avg_steps": float(results["steps"].mean()) if len(results) else 0.0, "new_samples": 2 * len(samples),
                "total_time": time.time() - t0, "num_workers": self.world, "mode": "hip", "buffer_size": len(self.buffer)}

    def train_network(self) -> dict:
        stats = {}
        if self.rank == 0:
            stats = training.train_network(self.current_model, self.optimizer, self.scheduler, self.buffer, self.config)
        if self.world > 1:
            xdist.broadcast_weights(self.current_model, src=0, device=self.device)
        return stats

    def evaluate(self) -> dict:
        stats = arena.evaluate_models(self.current_model, self.best_model, self.config, self.device, self.evaluator_kind)
        stats.pop("games", None)
        if stats["model_updated"]:
            self.best_model.load_state_dict(self.current_model.state_dict())
        else:
            self.current_model.load_state_dict(self.best_model.state_dict())
        return stats

    # ---- train.py:581-638 ------------------------------------------------------------------------------------
    def train(self, num_iterations: Optional[int] = None) -> list:
        cfg = self.config
        last = num_iterations if num_iterations is not None else cfg.num_iterations
        for iteration in range(self.iteration + 1, last + 1):
            self.iteration = iteration
            t0 = time.time()
            sp = self.self_play()
            tr = self.train_network()
            ev = {}
            if iteration % 2 == 0 and len(self.buffer) >= cfg.min_buffer_size:
                ev = self.evaluate()                   # deterministic: every rank reaches the same verdict
            if iteration % cfg.save_interval == 0 and self.rank == 0:
                training.save_checkpoint(cfg.checkpoint_dir, iteration, self.current_model, self.best_model,
                                         self.optimizer, self.scheduler, self.total_games, is_best=True)
            self.training_stats.append({"iteration": iteration, "time": time.time() - t0, "self_play": sp,
                                        "training": tr, "evaluation": ev})
            if self.rank == 0:
                os.makedirs(cfg.checkpoint_dir, exist_ok=True)
                with open(os.path.join(cfg.checkpoint_dir, "training_stats.json"), "w") as f:
                    json.dump(self.training_stats, f, indent=2, default=str)
        return self.training_stats
