"""The outer training loop on top of the self-play path -- `AlphaZeroTrainer.train()` (training/train.py:581-638) with
the engine underneath, one process per GPU (BASELINE configs[4]).

Per iteration, exactly the reference's order:
  1. self-play     every rank plays `shard_games(num_games_per_iter, world, rank)` games on its own engine with the
                   current best weights (no collective during search); the finished compact samples stay on the device
                   (`xq_engine_drain_device`) and one padded all-gather of those device buffers brings every rank's
                   records to every rank's device-resident replay buffer;
  2. train         `train_network` data-parallel over the ranks (every batch split, SyncBatchNorm, bucketed gradient
                   all-reduce: the reference's full-batch update) -- or on rank 0 alone with `config.ddp = False` -- then
                   ONE flat `broadcast_weights`;
  3. arena gate    every second iteration (train.py:609) the candidate plays the best model, the games sharded over the
                   ranks (`arena.evaluate_models`); promote at win_rate >= eval_win_rate, else the candidate reverts to
                   the best weights (train.py:525-533).  Every rank derives the verdict from the same all-reduced table
                   and rank 0's decision is broadcast on top, so the replicas cannot drift apart;
  4. checkpoint    every `save_interval` iterations and once more after the last one (train.py:613-615, 636-637), the
                   reference's file format; `training_stats.json` like train.py:620-634.
`resume(path)` is the reference's `--resume` (train.py:569-579, 759-761): iteration, total_games, both models, optimizer
and scheduler come back from a `checkpoint_iter*.pt` (written here or by the reference); the loop then continues at the
next iteration.  Like the reference the checkpoint itself does not hold the replay buffer; this loop additionally writes
`replay_buffer.pt` (its compact device records) next to it and restores it when present, so that a resumed run continues
exactly where an uninterrupted one would be.
Works unchanged with world_size 1, with or without a process group: every collective below runs whenever a group is
initialised (a one-rank RCCL group exercises the same calls as eight ranks).
"""
from __future__ import annotations

import copy
import json
import os
import time
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import arena, distributed as xdist, selfplay, training
from .model import XiangqiNet
from .sample_format import RESULT_DTYPE


class AlphaZeroLoop:
    def __init__(self, config, device="cuda", seed: int = 0, evaluator_kind: str = "hip"):
        self.config = config
        self.device = torch.device(device)
        self.grouped = dist.is_initialized()           # collectives run under ANY initialised group, one rank included
        self.world = dist.get_world_size() if self.grouped else 1
        self.rank = dist.get_rank() if self.grouped else 0
        self.seed = seed
        self.evaluator_kind = evaluator_kind
        torch.manual_seed(seed)                        # identical initial weights on every rank
        self.current_model = XiangqiNet(config.num_channels, config.num_res_blocks).to(self.device)
        self.best_model = copy.deepcopy(self.current_model)
        on_gpu = self.device.type == "cuda"
        from . import native_conv
        if on_gpu and native_conv.supported(config.num_channels):
            # train step on the hand-written kernels: tower convolutions (forward, data and weight gradient) and the fused
            # training-mode BatchNorm (native_conv.py); the deep copy above -- the self-play / arena model -- is not touched
            self.current_model.use_native_conv(True)
        # fused=True: the whole Adam update in one launch instead of torch's five foreach launches (same formula)
        self.optimizer = torch.optim.Adam(self.current_model.parameters(), lr=config.learning_rate,
                                          weight_decay=config.weight_decay, **({"fused": True} if on_gpu else {}))
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=config.lr_milestones,
                                                              gamma=config.lr_gamma)
        self.buffer = training.ReplayBuffer(config.max_buffer_size, self.device)
        self.iteration = 0
        self.total_games = 0
        self.training_stats = []

    # ---- the three stages --------------------------------------------------------------------------------
    def _play_shard(self, n_games: int):
        """-> (samples uint8[n,640], results uint8[m,16]) on self.device."""
        if n_games <= 0:
            return (torch.empty((0, 640), dtype=torch.uint8, device=self.device),
                    torch.empty((0, 16), dtype=torch.uint8, device=self.device))
        samples, results, _, _ = selfplay.run_games(self.best_model, self.config, n_games, self.device,
                                                    seed=self.seed + 1000 * self.iteration, rank=self.rank,
                                                    evaluator_kind=self.evaluator_kind, device_records=True)
        return samples, results

    def self_play(self) -> dict:
        cfg = self.config
        t0 = time.time()
        samples, results = self._play_shard(xdist.shard_games(cfg.num_games_per_iter, self.world, self.rank))
        if self.grouped:
            samples = xdist.all_gather_records_device(samples)
            results = xdist.all_gather_records_device(results)
        self.buffer.extend(samples)
        res = results.cpu().numpy().reshape(-1).view(RESULT_DTYPE) if len(results) else np.zeros(0, RESULT_DTYPE)
        self.total_games += len(res)
        wins = {1: 0, -1: 0, 0: 0}
        for r in res:
            wins[int(r["winner"])] += 1
        return {"games": len(res), "red_wins": wins[1], "black_wins": wins[-1], "draws": wins[0],
                "avg_steps": float(res["steps"].mean()) if len(res) else 0.0, "new_samples": 2 * int(samples.shape[0]),
                "total_time": time.time() - t0, "num_workers": self.world, "mode": "hip", "buffer_size": len(self.buffer)}

    def train_network(self) -> dict:
        """World 1: the reference's step.  World > 1: the same full-batch update computed data-parallel -- every rank
        holds the same buffer and weights, takes its slice of every batch, SyncBatchNorm + bucketed gradient all-reduce
        (`training.train_network(ddp=True)`); `config.ddp = False` falls back to training on rank 0 alone.  Either way one
        flat weight broadcast from rank 0 closes the step, so the replicas cannot drift."""
        stats = {}
        ddp = self.grouped and bool(getattr(self.config, "ddp", True)) and self.device.type == "cuda"
        # the batch order is a function of (seed, iteration): the same on every rank, and the same in a resumed run
        gen = torch.Generator().manual_seed(self.seed * 1000003 + 7919 * self.iteration + 17)
        if ddp or self.rank == 0:
            stats = training.train_network(self.current_model, self.optimizer, self.scheduler, self.buffer, self.config,
                                           generator=gen, ddp=ddp)
        if self.grouped:
            xdist.broadcast_weights(self.current_model, src=0, device=self.device)
        return stats

    def _arena(self) -> dict:
        return arena.evaluate_models(self.current_model, self.best_model, self.config, self.device, self.evaluator_kind)

    def evaluate(self) -> dict:
        stats = self._arena()
        stats.pop("games", None)
        if self.grouped:                               # one verdict for all replicas: rank 0's
            flag = torch.tensor([1 if stats["model_updated"] else 0], dtype=torch.int64,
                                device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            stats["model_updated"] = bool(flag.item())
        if stats["model_updated"]:
            self.best_model.load_state_dict(self.current_model.state_dict())
        else:
            self.current_model.load_state_dict(self.best_model.state_dict())
        return stats

    def _save(self, iteration: int) -> None:
        if self.rank == 0:
            training.save_checkpoint(self.config.checkpoint_dir, iteration, self.current_model, self.best_model,
                                     self.optimizer, self.scheduler, self.total_games, is_best=True)
            b = self.buffer                            # beside the reference's files: the compact replay records, oldest first
            oldest = (b.head - b.count) % b.cap
            rec = torch.roll(b.store, -oldest, 0)[:b.count].cpu()
            torch.save({"iteration": iteration, "records": rec}, os.path.join(self.config.checkpoint_dir, "replay_buffer.pt"))

    def resume(self, path: str) -> dict:
        """`AlphaZeroTrainer.load_checkpoint` + `--resume` (train.py:569-579, 759-761): restore iteration, total_games, both
        models, optimizer and scheduler from `path` (a `checkpoint_iter*.pt` of this loop or of the reference; loaded with
        `weights_only=True`); `train()` then continues with iteration + 1.  If `replay_buffer.pt` of the same iteration lies
        beside the checkpoint the replay buffer is restored too (the reference restarts with an empty one), and so is the
        `training_stats.json` history up to that iteration.  Every rank of a multi-rank run calls this with the same file."""
        info = training.load_checkpoint(path, self.current_model, self.best_model, self.optimizer, self.scheduler,
                                        map_location=self.device)
        self.iteration = int(info["iteration"])
        self.total_games = int(info["total_games"])
        folder = os.path.dirname(os.path.abspath(path))
        info["replay_buffer_restored"] = False
        rb = os.path.join(folder, "replay_buffer.pt")
        if os.path.exists(rb):
            saved = torch.load(rb, map_location="cpu", weights_only=True)
            if int(saved.get("iteration", -1)) == self.iteration:
                self.buffer = training.ReplayBuffer(self.config.max_buffer_size, self.device)
                self.buffer.extend(saved["records"])
                info["replay_buffer_restored"] = True
        st = os.path.join(folder, "training_stats.json")
        if os.path.exists(st):
            try:
                with open(st) as f:
                    self.training_stats = [e for e in json.load(f) if int(e.get("iteration", 0)) <= self.iteration]
            except (ValueError, OSError):
                self.training_stats = []
        return info

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
                ev = self.evaluate()
            if iteration % cfg.save_interval == 0:
                self._save(iteration)
            self.training_stats.append({"iteration": iteration, "time": time.time() - t0, "self_play": sp,
                                        "training": tr, "evaluation": ev})
            if self.rank == 0:
                os.makedirs(cfg.checkpoint_dir, exist_ok=True)
                with open(os.path.join(cfg.checkpoint_dir, "training_stats.json"), "w") as f:
                    json.dump(self.training_stats, f, indent=2, default=str)
        if self.iteration > 0:
            self._save(self.iteration)                 # train.py:636-637: the final state is always on disk
        return self.training_stats
