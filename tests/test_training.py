"""Next rows: train step on the device-resident compact buffer, checkpoint round trip."""
import json
import os

import numpy as np
import pytest

import golden_io as G
from test_host_logic import _oracle_game_as_compact

REF = "/root/reference/training"


def _trace():
    return json.load(open(os.path.join(G.GOLDEN, "train_trace.json")))


@pytest.mark.gpu
def test_batch_materialisation_matches_dense_adapter():
    """xq_samples_to_batch == the reference-schema adapter (planes, dense pi incl. mirror, z) for every logical sample."""
    import torch
    from xiangqi_alphazero_amd import training
    from xiangqi_alphazero_amd.sample_format import to_reference_tuples
    for t in G.game_traces()[:2]:
        arr, res = _oracle_game_as_compact(t)
        data, _ = to_reference_tuples(arr, res, augment=True)
        buf = training.ReplayBuffer(50000)
        buf.extend(arr)
        assert len(buf) == len(data)
        states, pi, z = buf.batch(torch.arange(len(data)))
        states, pi, z = states.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
        for j, (s, p, zz) in enumerate(data):
            np.testing.assert_array_equal(states[j], s)
            assert z[j, 0] == zz
            assert np.array_equal(np.nonzero(pi[j])[0], np.nonzero(p)[0])
            np.testing.assert_allclose(pi[j], p.astype(np.float32), rtol=2e-7, atol=0)


@pytest.mark.gpu
def test_replay_buffer_is_fifo_like_the_reference_deque():
    import torch
    from xiangqi_alphazero_amd import training
    arr, _ = _oracle_game_as_compact(G.game_traces()[1])
    buf = training.ReplayBuffer(max_size=20)            # 10 records
    buf.extend(arr[:7]); buf.extend(arr[7:16])
    assert len(buf) == 20
    st, _, _ = buf.batch(torch.tensor([0, 18]))
    from xiangqi_alphazero_amd.sample_format import encode_planes
    np.testing.assert_array_equal(st[0].cpu().numpy(), encode_planes(arr[6]["board"], int(arr[6]["side"])))
    np.testing.assert_array_equal(st[1].cpu().numpy(), encode_planes(arr[15]["board"], int(arr[15]["side"])))


@pytest.mark.gpu
def test_train_network_matches_reference_trace():
    """Same data, weights, optimiser and batch order as the reference's train_network run recorded in the fixture:
    losses within 1e-4 relative; probed weights within 1e-4 absolute -- they move by up to 1.2e-2 in the six Adam steps
    (lr 2e-3) and Adam's normalised update amplifies last-bit gradient differences between GPU and CPU kernels."""
    import types
    import torch
    from xiangqi_alphazero_amd import model, training, weights
    t = _trace()
    game = [x for x in G.game_traces() if x["name"] == t["game"]][0]
    arr, _ = _oracle_game_as_compact(game)
    buf = training.ReplayBuffer(50000)
    buf.extend(arr)
    assert len(buf) == t["n_samples"]
    net = model.XiangqiNet(*t["net"])
    net.load_state_dict(weights.make_state_dict(*t["net"], seed=t["seed"]))
    net = net.cuda()
    opt = torch.optim.Adam(net.parameters(), lr=t["lr"], weight_decay=t["weight_decay"])
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=t["milestones"], gamma=t["gamma"])
    cfg = types.SimpleNamespace(min_buffer_size=10, num_epochs=t["num_epochs"], batch_size=t["batch_size"])
    stats = training.train_network(net, opt, sch, buf, cfg, shuffle=False)
    for k in ("policy_loss", "value_loss", "total_loss", "learning_rate"):
        assert abs(stats[k] - t["stats"][k]) <= 1e-4 * abs(t["stats"][k]) + 1e-7, (k, stats[k], t["stats"][k])
    sd = net.state_dict()
    assert int(sd["input_conv.1.num_batches_tracked"]) == t["num_batches_tracked"]
    for k, want in t["probe"].items():
        np.testing.assert_allclose(sd[k].flatten()[:8].double().cpu().numpy(), want, rtol=0, atol=1e-4, err_msg=k)
    assert training.train_network(net, opt, sch, training.ReplayBuffer(100), cfg) == {}     # below min_buffer_size


def test_checkpoint_files_have_the_reference_layout(tmp_path):
    import torch
    from xiangqi_alphazero_amd import model, training, weights
    cur, best = model.XiangqiNet(16, 1), model.XiangqiNet(16, 1)
    cur.load_state_dict(weights.make_state_dict(16, 1, seed=1)); best.load_state_dict(weights.make_state_dict(16, 1, seed=2))
    opt = torch.optim.Adam(cur.parameters(), lr=0.002, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50, 80], gamma=0.1)
    path = training.save_checkpoint(str(tmp_path), 7, cur, best, opt, sch, total_games=123, is_best=True)
    ck = torch.load(path, weights_only=True)
    assert set(ck) == {"iteration", "model_state_dict", "best_model_state_dict", "optimizer_state_dict",
                       "scheduler_state_dict", "config", "total_games"}                      # train.py:539-550
    bm = torch.load(os.path.join(tmp_path, "best_model.pt"), weights_only=True)
    assert set(bm) == {"model_state_dict", "config", "iteration", "total_games"} and bm["config"] == {"num_channels": 16, "num_res_blocks": 1}
    c2, b2 = model.XiangqiNet(16, 1), model.XiangqiNet(16, 1)
    info = training.load_checkpoint(path, c2, b2, torch.optim.Adam(c2.parameters()), None)
    assert info["iteration"] == 7 and info["total_games"] == 123
    assert all(torch.equal(a, b) for a, b in zip(c2.state_dict().values(), cur.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(b2.state_dict().values(), best.state_dict().values()))
    assert torch.equal(model.load_reference_checkpoint(os.path.join(tmp_path, "best_model.pt")).state_dict()["policy_head.4.bias"],
                       best.state_dict()["policy_head.4.bias"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_checkpoint_loads_with_the_reference_loader(tmp_path):
    """A file written here is read by the reference's own AlphaZeroTrainer.load_checkpoint (train.py:569-579)."""
    import sys
    import types
    import torch
    from xiangqi_alphazero_amd import model, training, weights
    sys.dont_write_bytecode = True
    cur, best = model.XiangqiNet(16, 1), model.XiangqiNet(16, 1)
    cur.load_state_dict(weights.make_state_dict(16, 1, seed=5)); best.load_state_dict(weights.make_state_dict(16, 1, seed=6))
    opt = torch.optim.Adam(cur.parameters(), lr=0.002, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50, 80], gamma=0.1)
    path = training.save_checkpoint(str(tmp_path), 3, cur, best, opt, sch, total_games=40)
    sys.path[:0] = [REF, os.path.join(G.GOLDEN, "..", "..", "oracle", "_ref")]
    cwd = os.getcwd()
    os.chdir(str(tmp_path))                          # the reference opens 'training.log' in the cwd at import
    try:
        import train as ref_train
        import model as ref_model
    finally:
        os.chdir(cwd)
    rc, rb = ref_model.XiangqiNet(num_channels=16, num_res_blocks=1), ref_model.XiangqiNet(num_channels=16, num_res_blocks=1)
    ro = torch.optim.Adam(rc.parameters(), lr=0.002, weight_decay=1e-4)
    fake = types.SimpleNamespace(device="cpu", current_model=rc, best_model=rb, optimizer=ro,
                                 scheduler=torch.optim.lr_scheduler.MultiStepLR(ro, milestones=[50, 80], gamma=0.1))
    ref_train.AlphaZeroTrainer.load_checkpoint(fake, path)
    assert fake.iteration == 3 and fake.total_games == 40
    assert all(torch.equal(a, b) for a, b in zip(rc.state_dict().values(), cur.state_dict().values()))


@pytest.mark.gpu
def test_full_loop_two_iterations(tmp_path):
    """AlphaZeroTrainer.train() order on the engine: self-play -> train -> arena gate (iteration 2) -> checkpoint."""
    import types
    import torch
    from xiangqi_alphazero_amd import train_loop
    cfg = types.SimpleNamespace(
        num_channels=64, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=16,
        max_game_length=30, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True, random_opening_moves=4,
        num_iterations=2, batch_size=64, num_epochs=1, learning_rate=0.002, weight_decay=1e-4, lr_milestones=[50, 80],
        lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=100, eval_games=4, eval_win_rate=0.55, eval_simulations=8,
        checkpoint_dir=str(tmp_path), save_interval=2)
    loop = train_loop.AlphaZeroLoop(cfg, "cuda", seed=3)
    stats = loop.train()
    assert [s["iteration"] for s in stats] == [1, 2]
    assert stats[0]["self_play"]["games"] == 16 and stats[0]["self_play"]["mode"] == "hip"
    assert stats[0]["training"]["policy_loss"] > 0 and stats[1]["training"]["policy_loss"] < stats[0]["training"]["policy_loss"] + 1.0
    assert stats[0]["evaluation"] == {} and set(stats[1]["evaluation"]) == {"new_wins", "old_wins", "draws", "win_rate", "model_updated"}
    assert os.path.exists(tmp_path / "checkpoint_iter2.pt") and os.path.exists(tmp_path / "best_model.pt")
    saved = json.load(open(tmp_path / "training_stats.json"))
    assert len(saved) == 2 and saved[1]["self_play"]["buffer_size"] == len(loop.buffer)
    ev = stats[1]["evaluation"]
    same = all(torch.equal(a, b) for a, b in zip(loop.current_model.state_dict().values(), loop.best_model.state_dict().values()))
    assert same                                        # promoted or reverted: both leave current == best (train.py:525-533)
    assert ev["new_wins"] + ev["old_wins"] + ev["draws"] == 4


def _loop_rank_gpu(rank, world, port, tmp):
    import hashlib
    import types
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xiangqi_alphazero_amd import train_loop
    cfg = types.SimpleNamespace(
        num_channels=64, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=12,
        max_game_length=30, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True, random_opening_moves=4,
        num_iterations=2, batch_size=64, num_epochs=1, learning_rate=0.002, weight_decay=1e-4, lr_milestones=[50, 80],
        lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=100, eval_games=5, eval_win_rate=0.55, eval_simulations=8,
        checkpoint_dir=os.path.join(tmp, "ck%d" % rank), save_interval=2)
    loop = train_loop.AlphaZeroLoop(cfg, "cuda", seed=3)
    stats = loop.train()
    flat = torch.cat([t.reshape(-1).double() for t in list(loop.current_model.state_dict().values())
                      + list(loop.best_model.state_dict().values())]).cpu().numpy()
    digest = hashlib.sha256(flat.tobytes() + loop.buffer.store.cpu().numpy().tobytes()).hexdigest()
    ev = stats[1]["evaluation"]
    ok = (stats[0]["self_play"]["games"] == 12 and stats[0]["self_play"]["num_workers"] == 2 and len(loop.buffer) > 100
          and stats[1]["self_play"]["buffer_size"] == len(loop.buffer) and ev["new_wins"] + ev["old_wins"] + ev["draws"] == 5
          and (rank != 0 or stats[0]["training"]["policy_loss"] > 0))
    open(os.path.join(tmp, "gloop%d" % rank), "w").write("%d %s %d" % (int(bool(ok)), digest, len(loop.buffer)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_full_loop_two_ranks_share_one_gpu(tmp_path):
    """BASELINE configs[4]'s loop with world_size 2 (gloo; both ranks on the one GPU of the test box): games and arena
    games sharded, device-resident samples all-gathered, rank-0 training + broadcast.  Both ranks must end with
    bit-identical weights (current and best) and replay buffers."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_loop_rank_gpu, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [open(tmp_path / ("gloop%d" % r)).read().split() for r in range(2)]
    assert out[0][0] == "1" and out[1][0] == "1", out
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


def _ddp_rank(rank, world, port, tmp):
    import types
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xiangqi_alphazero_amd import model, training, weights
    t = _trace()
    game = [x for x in G.game_traces() if x["name"] == t["game"]][0]
    arr, _ = _oracle_game_as_compact(game)
    buf = training.ReplayBuffer(50000)
    buf.extend(arr)
    net = model.XiangqiNet(*t["net"])
    net.load_state_dict(weights.make_state_dict(*t["net"], seed=t["seed"]))
    net = net.cuda()
    opt = torch.optim.Adam(net.parameters(), lr=t["lr"], weight_decay=t["weight_decay"])
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=t["milestones"], gamma=t["gamma"])
    cfg = types.SimpleNamespace(min_buffer_size=10, num_epochs=t["num_epochs"], batch_size=t["batch_size"])
    stats = training.train_network(net, opt, sch, buf, cfg, shuffle=False, ddp=True)
    sd = net.state_dict()
    out = {"stats": stats, "probe": {k: [float(x) for x in sd[k].flatten()[:8].double().cpu()] for k in t["probe"]},
           "nbt": int(sd["input_conv.1.num_batches_tracked"]), "keys": list(sd.keys())}
    json.dump(out, open(os.path.join(tmp, "ddp%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_train_step_matches_reference_trace(tmp_path):
    """train.py:376-447 computed DATA-PARALLEL (world_size 2 over gloo, both ranks on the test box's one GPU): every batch
    split across the ranks, SyncBatchNorm, DDP gradient all-reduce.  Against the SAME recorded run of the reference's
    single-device train_network as the world-1 test: losses within 1e-4 relative, probed weights within 1e-4; both ranks
    end with the same weights; the state_dict keeps the reference's keys."""
    import socket
    import torch.multiprocessing as mp
    from xiangqi_alphazero_amd import weights
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t = _trace()
    outs = [json.load(open(tmp_path / ("ddp%d.json" % r))) for r in range(2)]
    for o in outs:
        for k in ("policy_loss", "value_loss", "total_loss", "learning_rate"):
            assert abs(o["stats"][k] - t["stats"][k]) <= 1e-4 * abs(t["stats"][k]) + 1e-7, (k, o["stats"][k], t["stats"][k])
        assert o["nbt"] == t["num_batches_tracked"]
        for k, want in t["probe"].items():
            np.testing.assert_allclose(o["probe"][k], want, rtol=0, atol=1e-4, err_msg=k)
        assert o["keys"] == list(weights.state_dict_shapes(*t["net"]).keys())
    assert outs[0]["probe"] == outs[1]["probe"]
