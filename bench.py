#!/usr/bin/env python3
"""bench.py -- MCTS simulations/s (+ measured self-play games/hour) of the MI355X self-play path.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts N fresh children (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torchrun would) before anything has touched the GPU,
waits for them and returns their exit status -- the reference fans its workers out by itself in the same way
(training/parallel_selfplay.py:284-293, 337-388).  Under an external launcher (torchrun) each rank runs main() directly.
A rank count that differs from --gpus is an error (non-zero exit), never a silently smaller run.

A *step* = one pass of the hot path over one batch: every one of the G resident games advances by ONE network
evaluation (k_select -> ResNet forward over [G,15,10,9] -> k_expand/backup).  Inputs (game state, trees, weights) are
resident in HBM when the timed region starts.  Simulations are counted by the engine itself (leaf evaluations and
terminal leaves; root evaluations are reported separately).

Workload at any N: BASELINE.json configs[2] per GPU -- 8192 concurrent games, 800 sims/move, 256ch x 10blk ResNet,
fp32, synthetic data (games from the opening + random opening plies, counter-generated weights).  Weak scaling: every
rank runs its own 8192 games; no collective in the data path.  The same line also carries
  * "peaked": the same measurement with peaked-policy weights (deep, narrow trees: SURVEY.md section 8d's second
    synthetic variant), unless --no-peaked;
  * "games_per_hour_measured": COMPLETE games under the reference's termination rules at BASELINE configs[1]
    (1024 games, 400 sims/move, 128x6) played inside this run, unless --complete-games 0;
  * "cpu_baseline" (N = 1, rank 0): the reference's mode-1 worker restated on the host cores;
  * "throughput_mode" (only with --throughput-mode): a labelled REDUCED-PRECISION second measurement (the hand-written bf16
    convolution, hip_net.HipBf16Evaluator) -- never the headline, outside the 1e-5 contract.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA = 157.3       # TFLOP/s, MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM = 8000.0           # GB/s


# FLOPs per evaluated position (2*MAC), SURVEY.md section 8a row a17
def net_flops(c, b):
    conv3 = 2 * 90 * 9
    tower = conv3 * 15 * c + b * 2 * conv3 * c * c
    heads = 2 * 90 * c * 32 + 2 * 2880 * 8100 + 2 * 90 * c * 4 + 2 * 360 * 128 + 2 * 128
    return tower + heads, tower


def workload_label(a):
    """BASELINE.json config this argument set is, derived from the arguments (never hard-coded)."""
    key = (a.games, a.sims, a.channels, a.blocks)
    names = {(1024, 400, 128, 6): "BASELINE configs[1]", (8192, 800, 256, 10): "BASELINE configs[2]",
             (8192, 800, 256, 20): "per-GPU share of BASELINE configs[3]"}
    base = names.get(key, "custom (not a BASELINE config)")
    return "%s: %d concurrent games per GPU, %d sims/move, %dch x %dblk ResNet%s" % (
        base, a.games, a.sims, a.channels, a.blocks, ", PEAKED policy weights (policy_gain 8)" if a.peaked else "")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--games", type=int, default=8192)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--evaluator", default="hip", choices=["hip", "torch", "nhwc", "bf16", "bf16-lib"])
    ap.add_argument("--throughput-mode", action="store_true",
                    help="add a second, labelled measurement with the reduced-precision evaluator (hand-written bf16 convolution)")
    ap.add_argument("--peaked", action="store_true", help="headline measurement itself on peaked-policy weights")
    ap.add_argument("--no-peaked", action="store_true", help="skip the second (peaked) measurement")
    ap.add_argument("--complete-games", type=int, default=1024,
                    help="games of the measured games/hour leg (BASELINE configs[1]; 0 disables)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 disables)")
    ap.add_argument("--graph", action="store_true",
                    help="time HIP-graph replays of the step (engine.capture_step) instead of eager launches; no per-stage breakdown")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-stagger", action="store_true",
                    help="all slots start their searches together (no start_stagger): with --prewarm P every slot is P simulations into "
                         "its first move when the timed steps start -- the counter passes use it with P = sims / 2, the mean of the "
                         "steady-state mix, because a counter pass over the full staggered prewarm does not finish on this pool")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher / rendezvous / reduction rehearsal WITHOUT the GPU path: every rank reports fixed stand-in "
                         "numbers through the same collectives and the line is labelled a rehearsal with value null (CPU test "
                         "of an 8-rank launch; never a measurement)")
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed network-free steps that bring the staggered slots to the steady-state mix of search "
                         "depths before warm-up (-1: sims+64, 0: off)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """Parent of a self-launched N-rank run.  Touches no GPU API; children are fresh interpreters."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), XQ_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                                  # one rank failed: stop exactly the children we started
                    q.terminate()
    return rc


def smi_sample(box):
    """One `rocm-smi` reading of the engine clock and the socket power WHILE the timed steps run (a child process; it does
    not touch the GPU from this one).  The boards are power-capped: under the fp32 MFMA tower the engine clock settles
    near 2.2 GHz of the nominal 2.4, which scales the MFMA peak that is actually available."""
    import re
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
        w = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        box["sclk_mhz"] = int(m.group(1)) if m else None
        box["power_w"] = float(w.group(1)) if w else None
    except Exception:
        box["sclk_mhz"] = box["power_w"] = None


class PseudoPolicy:
    """Network-free evaluator of the untimed prewarm: logits = gain * (planes . R) with a fixed seeded R, value 0.
    gain 0 -> uniform priors (shallow, wide trees); gain > 0 -> peaked priors (deep, narrow trees), so the slots reach the
    tree shape the timed steps will keep, at a few ms per step."""

    def __init__(self, games, device, gain):
        import torch
        self.gain = gain
        self.value = torch.zeros(games, dtype=torch.float32, device=device)
        if gain > 0:
            g = torch.Generator(device="cpu").manual_seed(99)
            self.r = (torch.randn(1350, 8100, generator=g) * gain).to(device)
        else:
            self.logits = torch.zeros((games, 8100), dtype=torch.float32, device=device)

    def __call__(self, x):
        if self.gain > 0:
            return x.reshape(x.shape[0], 1350) @ self.r, self.value
        return self.logits, self.value


def profiler_region(resume: bool):
    """XQ_BENCH_ROCTX=1 (set by tools/run_pmc_passes.sh): bracket the REAL steps with roctxProfilerResume / roctxProfilerPause so
    that `rocprofv3 --selected-regions --pmc ...` collects counters for them only -- a counter pass over the ~3 500 launches of
    the network-free prewarm does not finish on this pool.  No-op otherwise."""
    if os.environ.get("XQ_BENCH_ROCTX") != "1":
        return
    import ctypes
    lib = getattr(profiler_region, "_lib", None)
    if lib is None:
        lib = profiler_region._lib = ctypes.CDLL("librocprofiler-sdk-roctx.so")
        lib.roctxProfilerResume.argtypes = lib.roctxProfilerPause.argtypes = [ctypes.c_uint64]
    (lib.roctxProfilerResume if resume else lib.roctxProfilerPause)(0)


def measure(args, dev, rank, world, dist, backend, peaked):
    """One timed region: prewarm -> W warm-up steps -> barrier+sync, K steps, barrier+sync.  Returns the per-rank dict."""
    import torch
    from xiangqi_alphazero_amd import engine, evaluator, model, weights

    net = model.XiangqiNet(args.channels, args.blocks)
    net.load_state_dict(weights.make_state_dict(args.channels, args.blocks, policy_gain=8.0 if peaked else 1.0))
    ev, ev_name = evaluator.make_evaluator(net, dev, args.evaluator)
    cfg = engine.make_config(args.games, args.sims, max_game_length=400, random_opening_moves=8,
                             temperature_threshold=20, enable_resign=True, seed=args.seed, rank=rank,   # "full" preset
                             start_stagger=not args.no_stagger)
    eng = engine.SelfPlayEngine(cfg, dev, evaluator=ev)
    ev_t = [torch.cuda.Event(enable_timing=True) for _ in range(4 * args.steps)]
    sparse = hasattr(ev, "evaluate_legal")

    def sync():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():                  # any initialised group, a forced one-rank RCCL group included
            dist.barrier()
            torch.cuda.synchronize(dev)

    prewarm = args.sims + 64 if args.prewarm < 0 else args.prewarm
    if prewarm:
        # network-free, but through the SAME hand-off as the timed steps: under the sparse protocol the pseudo-policy's logits
        # are gathered at the ordered legal moves the engine asks for ([G,128] -> xq_engine_expand_legal), so every k_select /
        # k_expand launch of a run -- prewarm included -- is a product-path launch (the PMC passes average over all of them)
        pp = PseudoPolicy(args.games, dev, 1.2 if peaked else 0.0)
        zero_legal = torch.zeros((args.games, 128), dtype=torch.float32, device=dev) if sparse else None
        for _ in range(prewarm):
            lg, vl = pp(eng.select())
            if not sparse:
                eng.expand(lg, vl, False)
            elif pp.gain > 0:
                eng.expand_legal(torch.gather(lg, 1, (eng.req_moves.to(torch.int64) & 0xFFFF).clamp_(max=8099)), vl)   # entries past a row's count are ignored
            else:
                eng.expand_legal(zero_legal, vl)
        torch.cuda.synchronize(dev)
        del pp
    profiler_region(True)
    for _ in range(args.warmup):
        eng.step()
    sync()
    s0 = eng.stats()
    import threading
    smi = {}
    # not under a profiler: its preloaded library has the GPU initialised in every child before that child execs rocm-smi
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    smi_thread = threading.Thread(target=smi_sample, args=(smi,)) if (rank == 0 and not profiled) else None
    graphed = bool(args.graph and sparse and eng.capture_step(warmup=0))
    if hasattr(ev, "timing") and not graphed:
        ev.timing = True            # HIP events around each launch of the dominant kernel, timed region only
    sync()
    if smi_thread is not None:
        smi_thread.start()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if graphed:
            eng.step()
            continue
        e = ev_t[4 * k:4 * k + 4]
        e[0].record()
        x = eng.select()
        e[1].record()
        if sparse:                                 # logits of the legal moves only (xq_policy_head_legal), no dense row
            ll, value = ev.evaluate_legal(x, eng.req_moves, eng.req_counts)
            e[2].record()
            eng.expand_legal(ll, value)
        else:
            logits, value = ev(x)
            e[2].record()
            eng.expand(logits, value, False)
        e[3].record()
    sync()
    elapsed = time.perf_counter() - t0
    profiler_region(False)
    if smi_thread is not None:
        smi_thread.join()
    s1 = eng.stats()
    if hasattr(ev, "timing"):
        ev.timing = False
    n = args.steps
    if graphed:                     # one graph launch per step: no per-stage events; the whole step is booked as "evaluate"
        sel_ms, exp_ms, nn_ms = 1e-9, 1e-9, 1e3 * elapsed / n
    else:
        sel_ms = sum(ev_t[4 * k].elapsed_time(ev_t[4 * k + 1]) for k in range(n)) / n
        nn_ms = sum(ev_t[4 * k + 1].elapsed_time(ev_t[4 * k + 2]) for k in range(n)) / n
        exp_ms = sum(ev_t[4 * k + 2].elapsed_time(ev_t[4 * k + 3]) for k in range(n)) / n
    d = {k: s1[k] - s0[k] for k in ("sims", "depth_sum", "children_scanned", "nodes_created", "leaf_evals", "root_evals",
                                    "terminal_sims")}
    if hasattr(ev, "roofline") and graphed:
        # one launch per step, no per-kernel events: book the WHOLE step to the 2*blocks conv launches -- an upper bound on
        # the launch time, so `frac` is a lower bound here
        roof = ev.roofline(args.games, nn_ms, launch_ms=nn_ms / (2 * args.blocks))
        roof["kernel"] += " [graph replay: whole step / %d launches, lower bound]" % (2 * args.blocks)
    else:
        roof = ev.roofline(args.games, nn_ms) if hasattr(ev, "roofline") else None
    out = dict(elapsed=elapsed, sims=d["sims"], sel_ms=sel_ms, nn_ms=nn_ms, exp_ms=exp_ms, d=d, roof=roof, ev_name=ev_name,
               prewarm=prewarm, sparse=sparse, graphed=graphed, smi=smi)
    del eng, ev
    torch.cuda.empty_cache()
    return out


def reduce_ranks(m, dev, world, dist, backend):
    """-> (max elapsed over ranks, total sims, per-rank sims/s list)."""
    import torch
    if not dist.is_initialized():
        return m["elapsed"], float(m["sims"]), [m["sims"] / m["elapsed"]]
    tdev = dev if backend == "nccl" else "cpu"
    mine = torch.tensor([m["elapsed"], float(m["sims"])], dtype=torch.float64, device=tdev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    parts = [p.cpu() for p in parts]
    return max(float(p[0]) for p in parts), sum(float(p[1]) for p in parts), [float(p[1]) / float(p[0]) for p in parts]


def tree_block(args, m, peaked=False):
    d = m["d"]
    sims = max(d["sims"], 1)
    evals = max(d["leaf_evals"] + d["root_evals"], 1)
    tree = {"mean_depth": round(d["depth_sum"] / sims, 3), "children_read_per_sim": round(d["children_scanned"] / sims, 2),
            "children_created_per_eval": round(d["nodes_created"] / evals, 2), "root_evals": d["root_evals"],
            "terminal_sims": d["terminal_sims"]}
    # engine kernels against HBM: algorithmic bytes x units counted by the engine itself
    sel_bytes = 16 * d["children_scanned"] + 6 * d["depth_sum"] + 4 * (d["depth_sum"] + d["sims"]) + 5400 * evals + 2 * d["nodes_created"]
    # k_expand reads the legal moves' logits only (4 B each) under the sparse hand-off, the dense 32 400-B row otherwise
    exp_bytes = ((4 * d["nodes_created"]) if m["sparse"] else 32400 * evals) + 24 * d["nodes_created"] + 24 * (d["depth_sum"] + d["leaf_evals"])
    roof = {}
    for name, nbytes, ms in (("k_select", sel_bytes, m["sel_ms"]), ("k_expand", exp_bytes, m["exp_ms"])):
        gbs = nbytes / args.steps / (ms * 1e-3) / 1e9
        roof[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM, "unit": "GB/s", "frac": round(gbs / PEAK_HBM, 4),
                      "algorithmic_bytes_per_launch": int(nbytes / args.steps), "traffic": None}
    # HBM-side bytes per launch from the committed counter passes over the product path (tools/summarize_pmc_tree.py: real steps
    # only, request-size counters); counters cannot be read from inside this process, so GB/s here = those bytes x this run's
    # own launch time, and the ratio to the algorithmic bytes is the committed run's.
    if m["sparse"] and args.games == 8192 and args.sims == 800:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_tree_kernels.json")))
            v = [x for k, x in pmc["variants"].items() if k.startswith("peaked" if peaked else "near-uniform")][0]
            for name, ms in (("k_select", m["sel_ms"]), ("k_expand", m["exp_ms"])):
                t = v[name]["hbm_side_bytes_per_launch"]["total"]
                roof[name]["traffic"] = t
                roof[name]["traffic_over_algorithmic"] = v[name]["traffic_over_algorithmic"]
                roof[name]["counter_GBps_at_this_runs_launch_time"] = round(t / (ms * 1e-3) / 1e9, 1)
                roof[name]["traffic_source"] = "profiles/r03_pmc_tree_kernels.json (separate --pmc passes, sparse hand-off, real steps only)"
        except Exception:
            pass
    return tree, roof


def complete_games_leg(args, dev, rank, world, dist, backend):
    """Self-play games/hour MEASURED on complete games inside this run: BASELINE configs[1] (1024 concurrent games per
    GPU, 400 sims/move, 128x6), reference termination rules, games_target = games so the tail has idle slots (the figure
    under-states a refilling engine).  Timed like the main region: barrier + sync on both sides, max over ranks."""
    import numpy as np
    import torch
    from xiangqi_alphazero_amd import model, selfplay, weights

    class Cfg:                                  # "full" preset of training/train.py:692-704
        c_puct = 1.5; temperature_threshold = 20; max_game_length = 400; random_opening_moves = 8
        enable_resign = True; resign_threshold = -0.9; resign_check_steps = 5; num_simulations = 400

    net = model.XiangqiNet(128, 6)
    net.load_state_dict(weights.make_state_dict(128, 6))
    games = args.complete_games
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    samples, results, st, _ = selfplay.run_games(net, Cfg, games, dev, n_slots=min(games, 1024), seed=args.seed + 7, rank=rank,
                                                 evaluator_kind=args.evaluator, poll_every=256)
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    steps = np.array([int(r["steps"]) for r in results])
    mine = [elapsed, float(len(results)), float(st["sims"]), float(steps.sum())]
    if dist.is_initialized():
        tdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor(mine, dtype=torch.float64, device=tdev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        parts = [p.cpu().tolist() for p in parts]
    else:
        parts = [mine]
    wall = max(p[0] for p in parts)
    n_games = sum(p[1] for p in parts)
    return {"config": "BASELINE configs[1] per GPU: %d complete games, %d slots, 400 sims/move, 128x6" % (games, min(games, 1024)),
            "games_finished": int(n_games), "wall_s": round(wall, 2), "games_per_hour": round(n_games * 3600.0 / wall, 1),
            "simulations_per_s": round(sum(p[2] for p in parts) / wall, 1),
            "mean_plies_per_game": round(sum(p[3] for p in parts) / max(n_games, 1), 2),
            "launch": st.get("launch"), "slot_policy": "games_target = games: finished slots idle until the last game ends (tail-limited; "
                                                       "the refilling figure is tools/measure_games_per_hour.py --refill)",
            "refilling_engine": refill_reference(),
            "rank0": {"plies_p10": int(np.percentile(steps, 10)), "plies_p90": int(np.percentile(steps, 90)),
                      "red_wins": st["red_wins"], "black_wins": st["black_wins"], "draws": st["draws"], "samples": int(len(samples))}}


def refill_reference():
    """The steady-state figure of the same configuration with refilling slots, from the committed run of
    tools/measure_games_per_hour.py --refill (230 s of warm-up + 150 s: several game lengths, too long for a bench run).  NOT measured in
    this run, and labelled so."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_games_per_hour_refill_cfg1_1024slots_400sims_128x6.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return {"measured_in_this_run": False, "source": "profiles/" + os.path.basename(path), "games_per_hour": d["games_per_hour"],
                "simulations_per_s": d["simulations_per_s"], "per": "one GPU, 1024 slots, every finished slot starts a new game at once"}
    except (OSError, ValueError, KeyError):
        return None


def rehearse(args, rank, world, dist):
    """--rehearse: the launcher, the rendezvous and the reduction of an N-rank run with NO measurement behind them (gloo, no
    GPU API): rank r contributes elapsed 1 + r/100 s and 1000 (r + 1) simulations through the same all-gather as a real run.
    The line says so and carries no value."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    m = {"elapsed": 1.0 + rank / 100.0, "sims": 1000.0 * (rank + 1)}
    elapsed_max, sims_all, per_rank = reduce_ranks(m, "cpu", world, dist, "gloo")
    if rank == 0:
        print(json.dumps({
            "metric": "MCTS simulations/sec (whole node) + self-play games/hour, 256ch x 10blk ResNet", "value": None,
            "unit": "simulations/s", "n_gpus": world, "rehearsal": True,
            "note": "launcher / rendezvous / reduction rehearsal: stand-in per-rank numbers, nothing was measured",
            "ranks": {"launched_by": "bench.py" if os.environ.get("XQ_BENCH_SELF_LAUNCHED") else ("external launcher" if world > 1 else "single process"),
                      "backend": "gloo" if world > 1 else None, "world_size_seen": dist.get_world_size() if world > 1 else 1,
                      "elapsed_max_s": elapsed_max, "sims_total": sims_all,
                      "sims_per_s_min": round(min(per_rank), 1), "sims_per_s_max": round(max(per_rank), 1)}}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))            # nothing below runs in the launcher process

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a run of a different size" % (args.gpus, world))
    if args.rehearse:
        return rehearse(args, rank, world, dist)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one rank per GPU; XQ_BENCH_BACKEND=gloo lets several ranks share a card to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("XQ_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (one rank per GPU over RCCL)" % (world, n_dev))
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # XQ_BENCH_FORCE_GROUP=1: a process group even for ONE rank, so that a one-GPU box runs every collective of the N > 1
    # path (barriers, the all-gather of the per-rank figures) over RCCL
    if world > 1 or os.environ.get("XQ_BENCH_FORCE_GROUP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    m = measure(args, dev, rank, world, dist, backend, args.peaked)
    elapsed_max, sims_all, per_rank = reduce_ranks(m, dev, world, dist, backend)
    mp = None
    if not args.peaked and not args.no_peaked:
        mp = measure(args, dev, rank, world, dist, backend, True)
        mp_el, mp_sims, mp_rank = reduce_ranks(mp, dev, world, dist, backend)
    gph = complete_games_leg(args, dev, rank, world, dist, backend) if args.complete_games > 0 else None
    mt = None
    if args.throughput_mode and args.evaluator != "bf16":
        saved = args.evaluator
        args.evaluator = "bf16"
        mt = measure(args, dev, rank, world, dist, backend, args.peaked)
        mt_el, mt_sims, _ = reduce_ranks(mt, dev, world, dist, backend)
        args.evaluator = saved

    if rank == 0:
        flops_eval, _ = net_flops(args.channels, args.blocks)
        roof = m["roof"]
        not_measured = []
        if roof is not None and args.games == 8192 and args.channels == 256:
            # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE, separate runs of this same command): counters cannot be read from inside this process.
            for f in ("r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", f)))
                    fk = [v for k, v in pmc["FETCH_SIZE"].items() if "k_wino_conv" in k][0]["avg_KB_per_launch_raw"]
                    wk = [v for k, v in pmc["WRITE_SIZE"].items() if "k_wino_conv" in k][0]["avg_KB_per_launch_raw"]
                    roof["traffic"] = int((2 * fk + wk) * 1024)     # gfx950: FETCH_SIZE counts 16-B/lane reads at half
                    roof["traffic_source"] = "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE, bytes/launch; separate --pmc passes)" % f
                    not_measured.append("roofline.traffic")
                    break
                except Exception:
                    continue
        if roof is None:
            achieved = flops_eval * args.games / (m["nn_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "ResNet forward (%s), whole-network FLOPs / event-timed forward" % m["ev_name"],
                    "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA, 4), "traffic": None}
        if m["smi"].get("sclk_mhz") and roof.get("bound") == "mfma":
            # the peak the silicon offers at the clock the power cap left DURING the timed steps (nominal: 2400 MHz)
            mhz = m["smi"]["sclk_mhz"]
            roof["sclk_mhz_during_timed_steps"] = mhz
            roof["socket_power_w"] = m["smi"].get("power_w")
            roof["frac_of_peak_at_that_clock"] = round(roof["achieved"] / (PEAK_F32_MFMA * mhz / 2400.0), 4)
        tree, tree_roof = tree_block(args, m, args.peaked)
        value = sims_all / elapsed_max
        out = {
            "metric": "MCTS simulations/sec (whole node) + self-play games/hour, 256ch x 10blk ResNet",
            "value": round(value, 1), "unit": "simulations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed_max / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_label(args), "games_per_gpu": args.games, "sims_per_move": args.sims,
                       "net": "%dx%d" % (args.channels, args.blocks), "evaluator": m["ev_name"], "policy_handoff": "legal-move logits [G,128]" if m["sparse"] else "dense logits [G,8100]",
                       "launch": "HIP-graph replay, one launch per step" if m["graphed"] else "eager kernel launches",
                       "prewarm_steps": m["prewarm"], "slot_start": "together" if args.no_stagger else "staggered over one move",

                       "parallelism": "games sharded across ranks, no data-path collective"},
            "ranks": {"launched_by": "bench.py" if os.environ.get("XQ_BENCH_SELF_LAUNCHED") else ("external launcher" if world > 1 else "single process"),
                      "backend": backend if dist.is_initialized() else None, "world_size_seen": dist.get_world_size() if dist.is_initialized() else 1,
                      "sims_per_s_min": round(min(per_rank), 1), "sims_per_s_max": round(max(per_rank), 1)},
            "roofline": roof,
            "breakdown_ms": {"select": round(m["sel_ms"], 3), "evaluate": round(m["nn_ms"], 3), "expand_backup": round(m["exp_ms"], 3)},
            "tree": tree, "tree_roofline": tree_roof,
        }
        if mp is not None:
            ptree, ptree_roof = tree_block(args, mp, True)
            out["peaked"] = {"workload": "same configuration, peaked-policy weights (weights.make_state_dict policy_gain 8; prewarm "
                                         "under a peaked pseudo-policy): deep, narrow trees",
                             "value": round(mp_sims / mp_el, 1), "unit": "simulations/s", "ms_per_step": round(1e3 * mp_el / args.steps, 3),
                             "breakdown_ms": {"select": round(mp["sel_ms"], 3), "evaluate": round(mp["nn_ms"], 3),
                                              "expand_backup": round(mp["exp_ms"], 3)},
                             "tree": ptree, "tree_roofline": ptree_roof}
        if mt is not None:
            out["throughput_mode"] = {"label": "REDUCED PRECISION, not the headline and outside the 1e-5 contract: the residual tower on the "
                                               "hand-written bf16 Winograd convolution (hip_net.HipBf16Evaluator: transformed inputs and filters "
                                               "rounded to bf16, float32 accumulation, float32 activations in HBM, float32 stem and heads)",
                                      "roofline": mt["roof"],
                                      "dtype": "bf16", "value": round(mt_sims / mt_el, 1), "unit": "simulations/s",
                                      "ms_per_step": round(1e3 * mt_el / args.steps, 3), "evaluator": mt["ev_name"],
                                      "breakdown_ms": {"select": round(mt["sel_ms"], 3), "evaluate": round(mt["nn_ms"], 3),
                                                       "expand_backup": round(mt["exp_ms"], 3)}}
        # games/hour: measured on complete games at configs[1] (above); at this line's own configuration no game can finish
        # inside a bench run (one ply of all games = sims+1 steps), so that figure is DERIVED from the measured simulation
        # rate and the game length measured in this same run.
        if gph is not None:
            out["games_per_hour_measured"] = gph
            mpl = gph["mean_plies_per_game"]
            out["games_per_hour_derived"] = {"at_measured_mean_plies": round(value * 3600 / (args.sims * mpl), 1),
                                             "measured_mean_plies": mpl, "at_200_plies": round(value * 3600 / (args.sims * 200.0), 1)}
        else:
            out["games_per_hour_derived"] = {"at_200_plies": round(value * 3600 / (args.sims * 200.0), 1)}
        not_measured.append("games_per_hour_derived")
        out["not_measured_in_this_run"] = not_measured
        if args.cpu_seconds > 0 and world == 1:
            from oracle import cpu_baseline                      # the checker, timed beside the product path
            cb = cpu_baseline.run(args.channels, args.blocks, budget_s=args.cpu_seconds)
            out["cpu_baseline"] = {"value": round(cb["value"], 2), "unit": "simulations/s", "cores": cb["cores"], "kind": "port",
                                   "sample": "%d workers x one %d-simulation search from the opening, %dx%d fp32 batch-1 predict "
                                             "(%.1f ms), 1 thread each, %.1f s" % (cb["cores"], cb["sims_per_worker"], args.channels,
                                                                                 args.blocks, cb["predict_ms"], cb["seconds"])}
            if gph is not None:
                # the metric's second unit for the CPU side (sims/s, games/h, core count): the same
                # port at the games/hour leg's configuration (128x6, 400 sims/move), a bounded search sample, turned into
                # games/hour with the game length MEASURED on the complete games of this run (a complete CPU game at 400
                # sims/move is ~13 minutes per core: outside a bench run)
                cg = cpu_baseline.run(128, 6, budget_s=min(8.0, args.cpu_seconds))
                spg = 400.0 * gph["mean_plies_per_game"]
                out["cpu_baseline"]["games_per_hour"] = {
                    "value": round(cg["value"] * 3600.0 / spg, 2), "unit": "games/hour", "cores": cg["cores"], "kind": "port",
                    "simulations_per_s": round(cg["value"], 2),
                    "sample": "%d workers x one %d-simulation search, 128x6 fp32 batch-1 predict (%.1f ms), %.1f s; games/hour = "
                              "simulations/s x 3600 / (400 sims/move x %.1f plies per game measured on this run's complete games)"
                              % (cg["cores"], cg["sims_per_worker"], cg["predict_ms"], cg["seconds"], gph["mean_plies_per_game"]),
                    "gpu_over_cpu": round(gph["games_per_hour"] / max(cg["value"] * 3600.0 / spg, 1e-9), 1)}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
