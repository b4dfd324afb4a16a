"""Multi-GPU plumbing for the self-play path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

Games are independent units: rank r plays its share of the games on its own engine with its
own weight replica and tree arenas -- there is NO collective on the data path.  The only exchanges are
  * `broadcast_weights`  once per training iteration (one flat buffer; <= 188 MB fp32), replacing the reference's
    pickling of the whole state_dict to every worker (training/parallel_selfplay.py:339, 346-348);
  * `all_gather_samples` of finished compact samples (640 B each instead of the reference's 70 KB dense tuples that
    come back through `future.result()`, parallel_selfplay.py:376).  RCCL has no all-gatherv: counts are gathered
    first, blocks are padded to the largest count.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .sample_format import RESULT_DTYPE, SAMPLE_DTYPE


def shard_games(total_games: int, world_size: int, rank: int) -> int:
    """The reference's split rule (parallel_selfplay.py:291-293, 343): even shares, first `remainder` ranks +1."""
    base, rem = divmod(int(total_games), int(world_size))
    return base + (1 if rank < rem else 0)


def _flat_params(net: torch.nn.Module) -> List[torch.Tensor]:
    return [t for _, t in sorted(net.state_dict().items()) if t.is_floating_point()]


def broadcast_weights(net: torch.nn.Module, src: int = 0, device=None, group=None) -> None:
    """All floating-point entries of the state_dict travel as ONE flat buffer (one collective, not one per tensor)."""
    tensors = _flat_params(net)
    dev = device if device is not None else tensors[0].device
    flat = torch.cat([t.detach().reshape(-1).to(dev, torch.float32) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.device, t.dtype))
            off += n


def _all_gather_records(arr: np.ndarray, dtype: np.dtype, device, group) -> np.ndarray:
    world = dist.get_world_size(group)
    n_local = torch.tensor([len(arr)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    width = max(counts) if counts else 0
    if width == 0:
        return np.zeros(0, dtype=dtype)
    buf = torch.zeros(width * dtype.itemsize, dtype=torch.uint8, device=device)
    if len(arr):
        raw = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy())
        buf[:raw.numel()] = raw.to(device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = [p.cpu().numpy()[:c * dtype.itemsize].view(dtype) for p, c in zip(parts, counts)]
    return np.concatenate(out) if out else np.zeros(0, dtype=dtype)


def all_gather_samples(samples: np.ndarray, results: np.ndarray, device="cpu", group=None) -> Tuple[np.ndarray, np.ndarray]:
    """Every rank ends up with every rank's finished samples and game results (rank order)."""
    return (_all_gather_records(samples, SAMPLE_DTYPE, torch.device(device), group),
            _all_gather_records(results, RESULT_DTYPE, torch.device(device), group))


def all_gather_records_device(records: torch.Tensor, group=None) -> torch.Tensor:
    """uint8[n_local, width] on any device (GPU under RCCL, CPU under gloo) -> uint8[sum n, width] on the same device, rank
    order.  Counts are gathered first and every rank's block is padded to the largest (RCCL has no all-gatherv); the
    records never leave the device -- this is what carries the engine's compact samples (xq_engine_drain_device) to the
    trainer's device-resident replay buffer."""
    world = dist.get_world_size(group)
    dev = records.device
    width = records.shape[1]
    n_local = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    most = max(counts)
    if most == 0:
        return records[:0]
    buf = torch.zeros((most, width), dtype=torch.uint8, device=dev)
    buf[:records.shape[0]] = records
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])
