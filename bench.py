#!/usr/bin/env python3
"""bench.py -- MCTS simulations/s of the MI355X self-play path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* = one pass of the hot path over one batch: every one of the G resident games advances by ONE network
evaluation (k_select -> ResNet forward over [G,15,10,9] -> k_expand/backup).  Inputs (game state, trees, weights)
are resident in HBM when the timed region starts.  Simulations are counted by the engine itself (leaf evaluations
and terminal leaves; root evaluations are reported separately).

Workload at N=1: BASELINE.json configs[2] -- 8192 concurrent games, 800 sims/move, 256ch x 10blk ResNet, fp32,
synthetic data (games from the opening + random opening plies, counter-generated weights).  Weak scaling: every
rank runs its own 8192 games; no collective in the data path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# FLOPs per evaluated position (2*MAC), SURVEY.md section 8a row a17
def net_flops(c, b):
    conv3 = 2 * 90 * 9
    tower = conv3 * 15 * c + b * 2 * conv3 * c * c
    heads = 2 * 90 * c * 32 + 2 * 2880 * 8100 + 2 * 90 * c * 4 + 2 * 360 * 128 + 2 * 128
    return tower + heads, tower


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--games", type=int, default=8192)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--evaluator", default="auto", choices=["auto", "torch", "nhwc", "hip"])
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 disables)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed steps with a zero-logit evaluator that bring the staggered slots to the steady-state "
                         "mix of search depths before warm-up (-1: sims+64, 0: off)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one rank per GPU; XQ_BENCH_BACKEND=gloo lets several ranks share a card to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("XQ_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from xiangqi_alphazero_amd import engine, evaluator, model, weights

    net = model.XiangqiNet(args.channels, args.blocks)
    net.load_state_dict(weights.make_state_dict(args.channels, args.blocks))
    ev, ev_name = evaluator.make_evaluator(net, dev, args.evaluator)
    cfg = engine.make_config(args.games, args.sims, max_game_length=400, random_opening_moves=8,
                             temperature_threshold=20, enable_resign=True, seed=args.seed, rank=rank,   # "full" preset
                             start_stagger=True)
    eng = engine.SelfPlayEngine(cfg, dev, evaluator=ev)

    ev_t = [torch.cuda.Event(enable_timing=True) for _ in range(4 * args.steps)]

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # steady state: slots start staggered over one move's worth of steps; a zero-logit evaluator
    # (uniform priors, value 0 -- no network) advances them until every slot is somewhere inside a search.
    prewarm = args.sims + 64 if args.prewarm < 0 else args.prewarm
    if prewarm:
        z_logits = torch.zeros((args.games, 8100), dtype=torch.float32, device=dev)
        z_value = torch.zeros(args.games, dtype=torch.float32, device=dev)
        for _ in range(prewarm):
            eng.select()
            eng.expand(z_logits, z_value, False)
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        eng.step()
    sync()
    s0 = eng.stats()
    if hasattr(ev, "timing"):
        ev.timing = True            # HIP events around each launch of the dominant kernel, timed region only
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        e = ev_t[4 * k:4 * k + 4]
        e[0].record()
        x = eng.select()
        e[1].record()
        logits, value = ev(x)
        e[2].record()
        eng.expand(logits, value, False)
        e[3].record()
    sync()
    elapsed = time.perf_counter() - t0
    s1 = eng.stats()

    sel_ms = sum(ev_t[4 * k].elapsed_time(ev_t[4 * k + 1]) for k in range(args.steps)) / args.steps
    nn_ms = sum(ev_t[4 * k + 1].elapsed_time(ev_t[4 * k + 2]) for k in range(args.steps)) / args.steps
    exp_ms = sum(ev_t[4 * k + 2].elapsed_time(ev_t[4 * k + 3]) for k in range(args.steps)) / args.steps

    sims = s1["sims"] - s0["sims"]
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    t_sims = torch.tensor([float(sims)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_sims, op=dist.ReduceOp.SUM)
    elapsed_max, sims_all = float(t_el.item()), float(t_sims.item())

    if rank == 0:
        flops_eval, flops_tower = net_flops(args.channels, args.blocks)
        roof = ev.roofline(args.games, nn_ms) if hasattr(ev, "roofline") else None
        if roof is not None and args.games == 8192 and args.channels == 256:
            # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE, separate runs of this same command): counters cannot be read from inside this process.
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                fk = [v for k, v in pmc["FETCH_SIZE"].items() if "k_wino_conv" in k][0]["avg_KB_per_launch_raw"]
                wk = [v for k, v in pmc["WRITE_SIZE"].items() if "k_wino_conv" in k][0]["avg_KB_per_launch_raw"]
                roof["traffic"] = int((2 * fk + wk) * 1024)     # gfx950: FETCH_SIZE counts 16-B/lane reads at half
                roof["traffic_source"] = "profiles/r01_pmc_hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
            except Exception:
                pass
        if roof is None:
            achieved = flops_eval * args.games / (nn_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "ResNet forward (%s), whole-network FLOPs / event-timed forward" % ev_name,
                    "achieved": round(achieved, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 4),
                    "traffic": None}
        d_depth = s1["depth_sum"] - s0["depth_sum"]
        d_scan = s1["children_scanned"] - s0["children_scanned"]
        d_nodes = s1["nodes_created"] - s0["nodes_created"]
        out = {
            "metric": "MCTS simulations/sec (whole node) + self-play games/hour, 256ch x 10blk ResNet",
            "value": round(sims_all / elapsed_max, 1), "unit": "simulations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed_max / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d concurrent games per GPU, %d sims/move, %dch x %dblk ResNet"
                       % (args.games, args.sims, args.channels, args.blocks),
                       "games_per_gpu": args.games, "sims_per_move": args.sims, "net": "%dx%d" % (args.channels, args.blocks),
                       "evaluator": ev_name, "prewarm_steps": prewarm, "parallelism": "games sharded across ranks, no data-path collective"},
            "roofline": roof,
            "breakdown_ms": {"select": round(sel_ms, 3), "evaluate": round(nn_ms, 3), "expand_backup": round(exp_ms, 3)},
            "tree": {"mean_depth": round(d_depth / max(sims, 1), 3), "children_read_per_sim": round(d_scan / max(sims, 1), 2),
                     "children_created_per_eval": round(d_nodes / max(s1["leaf_evals"] + s1["root_evals"] - s0["leaf_evals"] - s0["root_evals"], 1), 2),
                     "root_evals": s1["root_evals"] - s0["root_evals"], "terminal_sims": s1["terminal_sims"] - s0["terminal_sims"]},
            "games_per_hour_derived": None,
        }
        # engine kernels against HBM: algorithmic bytes x units counted by the engine itself
        d_evals = s1["leaf_evals"] + s1["root_evals"] - s0["leaf_evals"] - s0["root_evals"]
        d_leaf = s1["leaf_evals"] - s0["leaf_evals"]
        sel_bytes = 16 * d_scan + 6 * d_depth + 4 * (d_depth + sims) + 5400 * d_evals + 2 * d_nodes
        exp_bytes = 32400 * d_evals + 24 * d_nodes + 24 * (d_depth + d_leaf)
        out["tree_roofline"] = {
            "k_select": {"bound": "hbm", "achieved": round(sel_bytes / args.steps / (sel_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(sel_bytes / args.steps / (sel_ms * 1e-3) / 8e12, 4)},
            "k_expand": {"bound": "hbm", "achieved": round(exp_bytes / args.steps / (exp_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(exp_bytes / args.steps / (exp_ms * 1e-3) / 8e12, 4)}}
        # games/hour cannot be observed in a few steps at 800 sims/move (one ply of all games = 801 steps); derive it
        # from the measured simulation rate with the reference's own game-length bound (<= 200 plies, game.py:595).
        out["games_per_hour_derived"] = {"at_200_plies": round(out["value"] * 3600 / (args.sims * 200.0), 1),
                                         "at_100_plies": round(out["value"] * 3600 / (args.sims * 100.0), 1)}
        try:    # mean game length MEASURED by tools/measure_games_per_hour.py (1024 complete games, configs[1])
            gp = json.load(open(os.path.join(ROOT, "profiles", "r01_games_per_hour_cfg1_1024x400_128x6.json")))
            mp = gp["plies_per_game"]["mean"]
            out["games_per_hour_derived"]["at_measured_mean_plies"] = round(out["value"] * 3600 / (args.sims * mp), 1)
            out["games_per_hour_derived"]["measured_mean_plies"] = mp
            out["games_per_hour_measured_configs1"] = {"games_per_hour": gp["games_per_hour"], "config": gp["config"]}
        except Exception:
            pass
        if args.cpu_seconds > 0 and world == 1:
            from oracle import cpu_baseline                      # the checker, timed beside the product path
            cb = cpu_baseline.run(args.channels, args.blocks, budget_s=args.cpu_seconds)
            out["cpu_baseline"] = {"value": round(cb["value"], 2), "unit": "simulations/s", "cores": cb["cores"], "kind": "port",
                                   "sample": "%d workers x one %d-simulation search from the opening, %dx%d fp32 batch-1 predict "
                                             "(%.1f ms), 1 thread each, %.1f s" % (cb["cores"], cb["sims_per_worker"], args.channels,
                                                                                 args.blocks, cb["predict_ms"], cb["seconds"])}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
