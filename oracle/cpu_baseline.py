"""CPU baseline leg of bench.py -- a PORT of the reference's mode-1 self-play worker
(training/parallel_selfplay.py:158-200, 337-388): W spawn'd processes, one thread each
(parallel_selfplay.py:160-165), own fp32 model copy, batch-1 `.predict` per simulation (training/mcts.py:143),
pointer-free C restatement of the search (oracle/xq_oracle.c).  TEST/BENCH INFRASTRUCTURE: never imported by the
product package; it is what the GPU path is compared WITH, not part of it.
"""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(args):
    channels, blocks, sims, seed, c_puct = args
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["MKL_NUM_THREADS"] = "1"
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from oracle import xq_oracle as O
    from xiangqi_alphazero_amd import model, weights

    net = model.XiangqiNet(channels, blocks)          # unfused eval-mode module, as the reference runs it
    net.load_state_dict(weights.make_state_dict(channels, blocks))
    net.eval()
    g = O.Game()
    rs = np.random.RandomState(seed)
    for a in range(seed % 5):                          # a few random plies so workers differ
        acts = g.legal_actions()
        g.make_action(int(acts[rs.randint(len(acts))]))
    net.predict(g.state_for_nn(), "cpu")               # warm-up (allocator, oneDNN primitive cache)
    noise = rs.dirichlet([0.3] * len(g.legal_actions()))
    t0 = time.perf_counter()
    res = O.mcts_search(g, sims, lambda s: net.predict(s, "cpu"), c_puct=c_puct, noise=noise)
    dt = time.perf_counter() - t0
    return sims, int(res.evals), dt


def host_cores() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box exposes all
    256 host CPUs to nproc but grants a 16-CPU share)."""
    n = os.cpu_count() or 2
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def time_one_predict(channels: int, blocks: int) -> float:
    import numpy as np
    import torch
    from xiangqi_alphazero_amd import model, weights
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        net = model.XiangqiNet(channels, blocks)
        net.load_state_dict(weights.make_state_dict(channels, blocks))
        x = np.zeros((15, 10, 9), dtype=np.float32)
        net.predict(x, "cpu")
        t0 = time.perf_counter()
        for _ in range(3):
            net.predict(x, "cpu")
        return (time.perf_counter() - t0) / 3
    finally:
        torch.set_num_threads(old)


def run(channels: int, blocks: int, budget_s: float = 20.0, workers: int | None = None, c_puct: float = 1.5):
    """-> dict(value sims/s, cores, sims_per_worker, seconds, evals).  Each worker runs ONE search of K
    simulations (K sized so the whole leg is ~budget_s of wall time); value = total sims / slowest worker."""
    import multiprocessing as mp
    cores = host_cores()
    if workers is None:
        workers = max(1, cores - 1)                     # the reference's rule, parallel_selfplay.py:287-288
    t_pred = time_one_predict(channels, blocks)
    sims = int(max(8, min(800, budget_s / max(2.0 * t_pred, 1e-4))))   # x2: co-running workers share memory bandwidth
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        out = pool.map(_worker, [(channels, blocks, sims, 100 + w, c_puct) for w in range(workers)])
    wall = time.perf_counter() - t0
    slowest = max(dt for _, _, dt in out)
    total = sum(s for s, _, _ in out)
    return dict(value=total / slowest, cores=workers, sims_per_worker=sims, seconds=slowest, wall_with_spawn=wall,
                evals=sum(e for _, e, _ in out), predict_ms=t_pred * 1e3)
