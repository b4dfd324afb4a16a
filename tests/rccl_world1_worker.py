"""Worker of tests/test_rccl_world1.py (a fresh process: the RCCL group is initialised before any other GPU work).

One rank, backend "nccl" (= RCCL on ROCm), device cuda:0: every collective of the multi-GPU path
runs once over RCCL on DEVICE tensors -- the flat weight broadcast, the padded device all-gather of compact records, the
arena's all-reduced winner table, the data-parallel train step (SyncBatchNorm + DistributedDataParallel's bucketed
all-reduce) and two iterations of AlphaZeroLoop with its broadcast verdict.  With one rank the collectives move no data
between GPUs, but they go through the same RCCL calls, device buffers and `get_backend() == "nccl"` branches as eight
ranks.  Prints one JSON line."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(tmp):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    from xiangqi_alphazero_amd import arena, distributed as xdist, model, train_loop, training, weights
    import golden_io as G
    from test_host_logic import _oracle_game_as_compact

    # 1. flat weight broadcast (device buffer)
    net = model.XiangqiNet(64, 1)
    net.load_state_dict(weights.make_state_dict(64, 1))
    net = net.to(dev)
    before = torch.cat([t.reshape(-1).float() for _, t in sorted(net.state_dict().items()) if t.is_floating_point()]).clone()
    xdist.broadcast_weights(net, src=0, device=dev)
    after = torch.cat([t.reshape(-1).float() for _, t in sorted(net.state_dict().items()) if t.is_floating_point()])
    out["broadcast_unchanged"] = bool(torch.equal(before, after))

    # 2. device all-gather of compact records (int64 counts + uint8 blocks on the device) and the host-array variant
    rec = torch.arange(5 * 640, dtype=torch.int64, device=dev).remainder(251).to(torch.uint8).view(5, 640)
    got = xdist.all_gather_records_device(rec)
    out["gather_device_ok"] = bool(got.is_cuda and torch.equal(got, rec))
    empty = xdist.all_gather_records_device(rec[:0])
    out["gather_empty_ok"] = bool(empty.shape[0] == 0)
    from xiangqi_alphazero_amd.sample_format import RESULT_DTYPE, SAMPLE_DTYPE
    smp = np.zeros(3, dtype=SAMPLE_DTYPE); smp["ply"] = np.arange(3)
    res = np.zeros(2, dtype=RESULT_DTYPE); res["winner"] = 1
    s2, r2 = xdist.all_gather_samples(smp, res, device=dev)
    out["gather_host_ok"] = bool(list(s2["ply"]) == [0, 1, 2] and list(r2["winner"]) == [1, 1])

    # 3. arena: sharded games (one shard), winner table all-reduced on the device
    cfg_a = types.SimpleNamespace(eval_games=3, eval_simulations=6, c_puct=1.5, max_game_length=12, eval_win_rate=0.55)
    other = model.XiangqiNet(64, 1)
    other.load_state_dict(weights.make_state_dict(64, 1, seed=7))
    ev = arena.evaluate_models(net, other.to(dev), cfg_a, dev)
    out["arena_games"] = int(ev["new_wins"] + ev["old_wins"] + ev["draws"])

    # 4. data-parallel train step over RCCL against the same step without a group's help (world 1: identical batches)
    t = json.load(open(os.path.join(G.GOLDEN, "train_trace.json")))
    game = [x for x in G.game_traces() if x["name"] == t["game"]][0]
    arr, _ = _oracle_game_as_compact(game)
    stats = {}
    for mode in ("ddp", "single"):
        buf = training.ReplayBuffer(50000, dev)
        buf.extend(arr)
        n2 = model.XiangqiNet(*t["net"])
        n2.load_state_dict(weights.make_state_dict(*t["net"], seed=t["seed"]))
        n2 = n2.to(dev)
        opt = torch.optim.Adam(n2.parameters(), lr=t["lr"], weight_decay=t["weight_decay"])
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=t["milestones"], gamma=t["gamma"])
        cfg_t = types.SimpleNamespace(min_buffer_size=10, num_epochs=t["num_epochs"], batch_size=t["batch_size"])
        stats[mode] = training.train_network(n2, opt, sch, buf, cfg_t, shuffle=False, ddp=(mode == "ddp"))
        if mode == "ddp":
            out["ddp_uses_syncbn"] = any(isinstance(m, torch.nn.SyncBatchNorm) for m in n2.modules())
            out["ddp_wrapper_cached"] = n2.__dict__.get("_xq_ddp") is not None
            stats["ddp_again"] = training.train_network(n2, opt, sch, buf, cfg_t, shuffle=False, ddp=True)   # same wrapper, no re-wrap
            out["ddp_wrapper_reused"] = n2.__dict__.get("_xq_ddp") is not None
    out["train_stats"] = stats
    out["reference_stats"] = t["stats"]

    # 5. two iterations of the outer loop under the group (second one runs the arena and its broadcast verdict)
    cfg = types.SimpleNamespace(
        num_channels=64, num_res_blocks=1, num_simulations=6, c_puct=1.5, temperature_threshold=10, num_games_per_iter=8,
        max_game_length=20, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True, random_opening_moves=4,
        num_iterations=2, batch_size=32, num_epochs=1, learning_rate=0.002, weight_decay=1e-4, lr_milestones=[50, 80],
        lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=50, eval_games=3, eval_win_rate=0.55, eval_simulations=6,
        checkpoint_dir=os.path.join(tmp, "ck"), save_interval=2)
    loop = train_loop.AlphaZeroLoop(cfg, dev, seed=3)
    st = loop.train()
    out["loop_iterations"] = [s["iteration"] for s in st]
    out["loop_games"] = [s["self_play"]["games"] for s in st]
    out["loop_trained"] = [bool(s["training"]) for s in st]
    out["loop_eval_keys"] = sorted(st[1]["evaluation"].keys())
    out["loop_grouped"] = bool(loop.grouped)
    out["loop_ddp_wrapper"] = loop.current_model.__dict__.get("_xq_ddp") is not None
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
