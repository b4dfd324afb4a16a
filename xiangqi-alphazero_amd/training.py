"""Train step and checkpoints -- the caller side of the self-play path.

`train_network` mirrors `AlphaZeroTrainer.train_network` (training/train.py:376-447): `num_epochs` passes over the whole
replay buffer in shuffled batches of `batch_size`, loss = -mean(sum(pi * log_softmax(logits))) + MSE(v, z), Adam,
gradient clipping at 1.0, one scheduler step per call, the same stats dict.  What differs is the data path: the
replay buffer holds the engine's compact 640-byte samples ON THE DEVICE (two logical samples per record: the position
and its mirror, in the reference's order s0, s0', s1, s1', ...), and every batch is materialised by one HIP kernel
(`xq_samples_to_batch`) instead of 70 KB/sample tuples going through a DataLoader.  The optimisation step itself is
torch autograd on the GPU (plumbing; <1 % of an iteration next to self-play).

`save_checkpoint` / `load_checkpoint` read and write the reference's files (train.py:537-579): `checkpoint_iter{N}.pt`
and `best_model.pt` with the same keys, so runs can move between the reference and this engine.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import hip
from .sample_format import SAMPLE_DTYPE


class ReplayBuffer:
    """FIFO of compact samples on the device; `max_size` counts LOGICAL samples like the reference's
    `deque(maxlen=max_buffer_size)` (train.py:206), i.e. max_size // 2 records."""

    def __init__(self, max_size: int = 50000, device="cuda", late_temperature: float = 0.3):
        self.device = torch.device(device)
        self.cap = max(1, max_size // 2)
        self.late_temperature = late_temperature
        self.store = torch.zeros((self.cap, hip.SAMPLE_BYTES), dtype=torch.uint8, device=self.device)
        self.count = 0          # records held
        self.head = 0           # next write position (ring)

    def __len__(self) -> int:
        return 2 * self.count

    def extend(self, samples) -> None:
        """Append finished samples, game order: a structured array of SAMPLE_DTYPE, or a uint8 tensor [n, 640] (device
        records of `SelfPlayEngine.drain_device` / `all_gather_records_device`: no host hop)."""
        if len(samples) == 0:
            return
        if isinstance(samples, torch.Tensor):
            raw = samples.view(-1, hip.SAMPLE_BYTES)
        else:
            raw = torch.from_numpy(np.ascontiguousarray(samples).view(np.uint8).reshape(len(samples), hip.SAMPLE_BYTES))
        raw = raw[-self.cap:].to(self.device)
        n = raw.shape[0]
        first = min(n, self.cap - self.head)
        self.store[self.head:self.head + first] = raw[:first]
        if n > first:
            self.store[:n - first] = raw[first:]
        self.head = (self.head + n) % self.cap
        self.count = min(self.cap, self.count + n)

    def _record_of(self, logical: torch.Tensor):
        """logical index (oldest first, s0, s0', s1, ...) -> (ring record index, flip flag)."""
        rec = torch.div(logical, 2, rounding_mode="floor")
        oldest = (self.head - self.count) % self.cap
        return ((rec + oldest) % self.cap).to(torch.int32), (logical % 2).to(torch.uint8)

    def batch(self, logical: torch.Tensor):
        """-> states f32[B,15,10,9], pi f32[B,8100], z f32[B,1] for the given logical indices (device int64)."""
        idx, flip = self._record_of(logical.to(self.device))
        b = idx.shape[0]
        states = torch.empty((b, 15, 10, 9), dtype=torch.float32, device=self.device)
        pi = torch.empty((b, hip.ACTION_SPACE), dtype=torch.float32, device=self.device)
        z = torch.empty((b, 1), dtype=torch.float32, device=self.device)
        hip.check(hip.lib().xq_samples_to_batch(self.store.data_ptr(), idx.contiguous().data_ptr(),
                                                flip.contiguous().data_ptr(), b, float(self.late_temperature),
                                                states.data_ptr(), pi.data_ptr(), z.data_ptr(),
                                                hip.stream_ptr(self.device)), "xq_samples_to_batch")
        self._keep = (idx, flip)
        return states, pi, z


def prepare_ddp(model, device, group=None):
    """The data-parallel form of `model`, built ONCE and kept for the model's lifetime: BatchNorm -> SyncBatchNorm IN PLACE
    (same Parameters and buffers, same state_dict keys: checkpoints, `broadcast_weights` and the evaluators are unaffected)
    and one DistributedDataParallel wrapper (its construction broadcasts the parameters, so it is not repeated per call).
    SIDE EFFECT on the caller's model: from here on a training-mode forward is a collective -- every rank of `group` must
    take part in it; `train_network(ddp=False)` refuses such a model under a multi-rank group (`revert_sync_batchnorm`
    undoes the conversion)."""
    wrapper = model.__dict__.get("_xq_ddp")
    if wrapper is None:
        torch.nn.SyncBatchNorm.convert_sync_batchnorm(model, group)       # in place for the children
        device = torch.device(device)
        wrapper = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[device.index] if device.type == "cuda" else None, process_group=group, broadcast_buffers=False)
        model.__dict__["_xq_ddp"] = wrapper       # not a registered submodule (the wrapper already holds the model)
    return wrapper


def revert_sync_batchnorm(model) -> None:
    """Undo `prepare_ddp`: every SyncBatchNorm becomes a BatchNorm2d again (same tensors) and the cached wrapper is dropped."""
    model.__dict__.pop("_xq_ddp", None)

    def walk(mod):
        for name, child in list(mod.named_children()):
            if isinstance(child, torch.nn.SyncBatchNorm):
                bn = torch.nn.BatchNorm2d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
                bn.weight, bn.bias = child.weight, child.bias
                bn.running_mean, bn.running_var, bn.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
                bn.train(child.training)
                setattr(mod, name, bn)
            else:
                walk(child)
    walk(model)


def train_network(model, optimizer, scheduler, buffer: ReplayBuffer, config, shuffle: bool = True,
                  generator: Optional[torch.Generator] = None, ddp: bool = False, group=None) -> Dict[str, float]:
    """One call of the reference's train_network (train.py:376-447) on the device-resident buffer.

    `ddp=True` under an initialised torch.distributed group (every rank holds the same buffer and the same weights, as
    AlphaZeroLoop keeps them): every batch is split across the ranks, BatchNorm statistics are synchronised (SyncBatchNorm:
    the batch statistics of the WHOLE batch, as on the reference's single device) and the gradients are summed by bucketed
    all-reduce overlapped with backward (DistributedDataParallel) -- the update is the reference's full-batch update, the
    replicas stay identical, and each GPU runs 1/world of the forward/backward work.  The conversion and the wrapper are
    made once per model (`prepare_ddp`, which documents the side effect on the caller's model).  The batch order comes
    from `generator` (or a generator seeded identically on every rank)."""
    if len(buffer) < config.min_buffer_size:
        return {}
    import torch.distributed as dist
    use_ddp = bool(ddp and dist.is_initialized())
    world = dist.get_world_size(group) if use_ddp else 1
    rank = dist.get_rank(group) if use_ddp else 0
    n = len(buffer)
    model.train()
    net = model
    if use_ddp:
        net = prepare_ddp(model, buffer.device, group)
        if generator is None:                                              # one order for all ranks
            generator = torch.Generator().manual_seed(int(scheduler.last_epoch) * 7919 + 17)
    elif dist.is_initialized() and dist.get_world_size(group) > 1 and any(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules()):
        # a training-mode forward of this model is a collective; taken by a subset of the ranks it would hang
        raise hip.XqError("train_network(ddp=False): the model still carries SyncBatchNorm from a ddp step and the process "
                          "group has %d ranks; train with ddp=True on every rank or call revert_sync_batchnorm(model) first"
                          % dist.get_world_size(group))
    # the two losses of every batch are added up ON THE DEVICE in float64 (the same additions, in the same order, as the reference's
    # Python floats, train.py:421-423) and read once at the end: no host synchronisation inside the epoch loop
    totals = None                                                      # float64[2] on the losses' device
    batches = 0
    for _ in range(config.num_epochs):
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        for lo in range(0, n, config.batch_size):                     # drop_last=False
            idx = order[lo:lo + config.batch_size]
            full = idx.numel()
            scale = 1.0
            if use_ddp and full >= world:                              # this rank's slice of the batch
                idx = idx.tensor_split(world)[rank]
                scale = float(world)                                   # DDP averages the ranks' gradients
            states, target_pi, target_z = buffer.batch(idx)
            logits, value = net(states)
            if use_ddp and full >= world:
                # sums over the local slice, scaled so that DDP's average over ranks is the full-batch mean loss
                policy_loss = -torch.sum(target_pi * F.log_softmax(logits, dim=1)) / full
                value_loss = torch.sum((value - target_z) ** 2) / full
                loss = (policy_loss + value_loss) * scale
            else:                                                      # the reference's expressions (train.py:409-413)
                policy_loss = -torch.mean(torch.sum(target_pi * F.log_softmax(logits, dim=1), dim=1))
                value_loss = F.mse_loss(value, target_z)
                loss = policy_loss + value_loss
            optimizer.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            optimizer.step()
            pv = torch.stack([policy_loss.detach(), value_loss.detach()])
            if use_ddp and full >= world:
                dist.all_reduce(pv, group=group)                       # the slices' shares add up to the batch's losses
            totals = pv.double() if totals is None else totals + pv.double()
            batches += 1
    scheduler.step()
    total_p, total_v = totals.tolist() if totals is not None else (0.0, 0.0)
    return {"policy_loss": total_p / max(batches, 1), "value_loss": total_v / max(batches, 1),
            "total_loss": (total_p + total_v) / max(batches, 1), "learning_rate": optimizer.param_groups[0]["lr"]}


def save_checkpoint(checkpoint_dir: str, iteration: int, current_model, best_model, optimizer, scheduler, total_games: int,
                    is_best: bool = False) -> str:
    """Writes the reference's two files with the reference's keys (train.py:537-567)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    cfg = {"num_channels": current_model.num_channels, "num_res_blocks": current_model.num_res_blocks}
    path = os.path.join(checkpoint_dir, f"checkpoint_iter{iteration}.pt")
    torch.save({"iteration": iteration, "model_state_dict": current_model.state_dict(),
                "best_model_state_dict": best_model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "scheduler_state_dict": scheduler.state_dict(), "config": cfg, "total_games": total_games}, path)
    if is_best:
        torch.save({"model_state_dict": best_model.state_dict(), "config": cfg, "iteration": iteration,
                    "total_games": total_games}, os.path.join(checkpoint_dir, "best_model.pt"))
    return path


def load_checkpoint(path: str, current_model, best_model, optimizer=None, scheduler=None, map_location="cpu") -> dict:
    """Reads a `checkpoint_iter*.pt` written by the reference or by save_checkpoint (train.py:569-579).
    `weights_only=True`: nothing in the file is executed."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    current_model.load_state_dict(ck["model_state_dict"])
    best_model.load_state_dict(ck["best_model_state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ck:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return {"iteration": ck["iteration"], "total_games": ck.get("total_games", 0), "config": ck.get("config", {})}
