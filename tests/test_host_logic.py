"""CPU tests of the host side of the boundary: compact -> dense sample adapter (against the reference's recorded
game and augmentation fixtures), and the multi-rank plumbing over gloo (world_size 2)."""
import os
import zlib

import numpy as np
import pytest

import golden_io as G
from draws import Draws
from oracle import xq_oracle as O
from stub_eval import StubEvaluator


def _oracle_game_as_compact(t):
    from xiangqi_alphazero_amd.sample_format import RESULT_DTYPE, SAMPLE_DTYPE
    d = Draws(t["seed"])
    ev = StubEvaluator(peaked=(t["stub"] == "peaked"))
    samples, winner, steps, _, _ = O.play_one_game(t["cfg"], ev.predict, d.randint, d.choice_index, d.dirichlet, d.uniform)
    arr = np.zeros(len(samples), dtype=SAMPLE_DTYPE)
    for i, s in enumerate(samples):
        n = len(s["actions"])
        arr[i]["board"] = s["board"]; arr[i]["side"] = s["player"]; arr[i]["z"] = s["z"]; arr[i]["n_moves"] = n
        arr[i]["late_temp"] = 0 if s["temperature"] == 1.0 else 1
        arr[i]["ply"] = i; arr[i]["slot"] = 3; arr[i]["game_seq"] = 1
        arr[i]["actions"][:n] = s["actions"]; arr[i]["visits"][:n] = s["visits"]
    res = np.zeros(1, dtype=RESULT_DTYPE)
    res[0]["slot"] = 3; res[0]["game_seq"] = 1; res[0]["winner"] = winner; res[0]["steps"] = steps
    res[0]["n_samples"] = len(samples)
    return arr, res


def test_dense_adapter_reproduces_reference_schema_and_augmentation():
    from xiangqi_alphazero_amd.sample_format import to_reference_tuples
    t = G.game_traces()[0]
    arr, res = _oracle_game_as_compact(t)
    data, per_game = to_reference_tuples(arr, res, augment=True)
    assert per_game == [(t["winner"], t["steps"], len(t["plies"]))] and len(data) == 2 * len(t["plies"])
    for i, want in enumerate(t["plies"]):
        state, pi, z = data[2 * i]
        assert state.dtype == np.float32 and state.shape == (15, 10, 9) and pi.dtype == np.float64 and pi.shape == (8100,)
        assert zlib.crc32(state.tobytes()) & 0xFFFFFFFF == want["state_crc"] and z == want["z"]
        nz = np.nonzero(pi)[0]
        assert list(nz) == want["pi"]["idx"]
        np.testing.assert_allclose(pi[nz], [G.hexf(x) for x in want["pi"]["val"]], rtol=1e-14, atol=0)
    # the reference's own _augment_data output for the first two samples (original, flipped, original, flipped)
    for j, want in enumerate(t["augmented_first2"]):
        state, pi, z = data[j]
        assert zlib.crc32(np.ascontiguousarray(state).tobytes()) & 0xFFFFFFFF == want["state_crc"] and z == want["z"]
        nz = np.nonzero(pi)[0]
        assert list(nz) == want["pi"]["idx"]
        np.testing.assert_allclose(pi[nz], [G.hexf(x) for x in want["pi"]["val"]], rtol=1e-14, atol=0)


def test_flip_perm_matches_reference():
    from xiangqi_alphazero_amd.sample_format import FLIP_PERM, encode_planes
    np.testing.assert_array_equal(FLIP_PERM, G.flip_perm())
    d = G.corpus()
    for i in range(0, len(d["board"]), 211):
        np.testing.assert_array_equal(encode_planes(d["board"][i], int(d["side"][i])),
                                      O.encode_state(d["board"][i], int(d["side"][i])))


def test_shard_games_rule():
    from xiangqi_alphazero_amd.distributed import shard_games
    for total, world in ((65536, 8), (50, 7), (3, 8), (20, 3)):
        shares = [shard_games(total, world, r) for r in range(world)]
        assert sum(shares) == total and max(shares) - min(shares) <= 1 and shares == sorted(shares, reverse=True)


def _rank_main(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xiangqi_alphazero_amd import distributed as D, model
    from xiangqi_alphazero_amd.sample_format import RESULT_DTYPE, SAMPLE_DTYPE
    # weights: rank 0's values must arrive everywhere
    torch.manual_seed(100 + rank)
    net = model.XiangqiNet(16, 1)
    D.broadcast_weights(net, src=0)
    ref = model.XiangqiNet(16, 1)
    torch.manual_seed(100)
    ref2 = model.XiangqiNet(16, 1)
    same = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref2.state_dict().values()))
    # samples: ragged counts (rank 1 has none of its own results)
    n = [5, 0][rank] if world == 2 else rank + 1
    smp = np.zeros(n, dtype=SAMPLE_DTYPE)
    smp["slot"] = rank; smp["ply"] = np.arange(n); smp["z"] = 1 - 2 * rank
    res = np.zeros(2 - rank, dtype=RESULT_DTYPE)
    res["slot"] = rank; res["winner"] = 1 - 2 * rank
    all_s, all_r = D.all_gather_samples(smp, res, device="cpu")
    ok = (len(all_s) == 5 and list(all_s["slot"]) == [0] * 5 and list(all_s["ply"]) == list(range(5))
          and len(all_r) == 3 and list(all_r["slot"]) == [0, 0, 1] and list(all_r["winner"]) == [1, 1, -1])
    open(os.path.join(tmp, "ok%d" % rank), "w").write("%d %d" % (int(same), int(ok)))
    dist.barrier()
    dist.destroy_process_group()
    del ref


def test_two_rank_gloo_broadcast_and_gather(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / ("ok%d" % r)).read() == "1 1"


def _loop_rank_main(rank, world, port, tmp):
    """AlphaZeroLoop's collectives and control flow over gloo on CPU tensors.  The three GPU stages are replaced by
    deterministic stand-ins (the engine, the HIP batch kernel and the arena games need the GPU; tests/test_training.py
    runs the real ones there): what is under test is the sharding, the device-record all-gather into every rank's replay
    buffer, rank-0 training + weight broadcast, the sharded arena table + broadcast verdict, and the final checkpoint."""
    import types
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xiangqi_alphazero_amd import arena, train_loop, training
    from xiangqi_alphazero_amd.sample_format import RESULT_DTYPE, SAMPLE_DTYPE
    cfg = types.SimpleNamespace(
        num_channels=16, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=5,
        max_game_length=30, random_opening_moves=2, enable_resign=False, resign_threshold=-0.9, resign_check_steps=5,
        learning_rate=0.01, weight_decay=1e-4, lr_milestones=[50], lr_gamma=0.1, max_buffer_size=64, min_buffer_size=4,
        num_epochs=1, batch_size=8, eval_games=7, eval_simulations=4, eval_win_rate=0.55, save_interval=5,
        num_iterations=3, checkpoint_dir=os.path.join(tmp, "ck%d" % rank))

    class Loop(train_loop.AlphaZeroLoop):
        def _play_shard(self, n_games):                 # n_games games of 3 samples each, tagged with rank and iteration
            smp = np.zeros(3 * n_games, dtype=SAMPLE_DTYPE)
            smp["slot"] = self.rank; smp["game_seq"] = self.iteration; smp["ply"] = np.arange(3 * n_games)
            res = np.zeros(n_games, dtype=RESULT_DTYPE)
            res["slot"] = self.rank; res["winner"] = 1 - 2 * self.rank; res["steps"] = 10 + self.rank
            return (torch.from_numpy(smp.view(np.uint8).reshape(-1, 640).copy()),
                    torch.from_numpy(res.view(np.uint8).reshape(-1, 16).copy()))

    def fake_train(model, optimizer, scheduler, buffer, config, **kw):   # only rank 0 is ever asked to train
        assert dist.get_rank() == 0
        with torch.no_grad():
            for p_ in model.parameters():
                p_.add_(0.01 * len(buffer))
        return {"policy_loss": 1.0, "value_loss": 0.5, "total_loss": 1.5, "learning_rate": 0.01}

    played = []

    def fake_play_arena(en, eo, n, sims, maxlen, c_puct, device, policy_is_probs=False, first_game=0):
        played.append((first_game, n))
        res = np.zeros(n, dtype=RESULT_DTYPE)
        games = np.arange(first_game, first_game + n)
        res["slot"] = np.arange(n)
        res["winner"] = np.where(games % 2 == 0, 1, np.where(games % 3 == 0, 0, -1))   # new (red in even games) wins a lot
        res["steps"] = 20 + games
        return res

    training.train_network = fake_train
    arena.play_arena = fake_play_arena
    arena.ev_mod.make_evaluator = lambda net, device, kind: (None, "fake")
    loop = Loop(cfg, device="cpu", seed=3)
    stats = loop.train()
    sd_c = torch.cat([t.reshape(-1).double() for t in loop.current_model.state_dict().values()])
    sd_b = torch.cat([t.reshape(-1).double() for t in loop.best_model.state_dict().values()])
    import hashlib
    digest = hashlib.sha256(sd_c.numpy().tobytes() + sd_b.numpy().tobytes() + loop.buffer.store.numpy().tobytes()).hexdigest()
    ev = stats[1]["evaluation"]
    shares = [3, 2]                                    # 5 games over 2 ranks
    ok = (len(stats) == 3 and stats[0]["self_play"]["games"] == 5 and stats[0]["self_play"]["new_samples"] == 30
          and loop.total_games == 15 and len(loop.buffer) == 2 * min(32, 45) and loop.buffer.count == 32
          and (ev["new_wins"], ev["old_wins"], ev["draws"]) == (6, 0, 1) and ev["model_updated"]
          and played == ([(0, 4)] if rank == 0 else [(4, 3)]))
    ok = ok and stats[0]["self_play"]["red_wins"] == shares[0] and stats[0]["self_play"]["black_wins"] == shares[1]
    ok = ok and (rank != 0 or (os.path.exists(os.path.join(cfg.checkpoint_dir, "checkpoint_iter3.pt"))
                               and os.path.exists(os.path.join(cfg.checkpoint_dir, "best_model.pt"))))
    open(os.path.join(tmp, "loop%d" % rank), "w").write("%d %s" % (int(bool(ok)), digest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_alphazero_loop(tmp_path):
    """world_size 2 over gloo: three iterations; both ranks must end with identical weights (current and best) and an
    identical replay buffer, and rank 0 must have written the final checkpoint (train.py:636-637)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_loop_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [open(tmp_path / ("loop%d" % r)).read().split() for r in range(2)]
    assert out[0][0] == "1" and out[1][0] == "1", out
    assert out[0][1] == out[1][1]


def test_alphazero_loop_resume_equals_uninterrupted(tmp_path):
    """`AlphaZeroLoop.resume` = the reference's --resume (training/train.py:569-579, 759-761): a loop stopped after
    iteration 2 and resumed in a NEW object from checkpoint_iter2.pt continues at iteration 3 and ends in exactly the state
    of an uninterrupted run -- weights (current and best), optimizer moments and step counts, scheduler epoch and learning
    rate, total_games, replay buffer and the training_stats history.  World size 1, CPU; the GPU stages are deterministic
    stand-ins (the real ones run in tests/test_training.py), the optimisation step is real Adam on a real loss."""
    import types
    import torch
    from xiangqi_alphazero_amd import train_loop, training
    from xiangqi_alphazero_amd.sample_format import RESULT_DTYPE, SAMPLE_DTYPE

    def cfg_for(name):
        return types.SimpleNamespace(
            num_channels=16, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=4,
            max_game_length=30, random_opening_moves=2, enable_resign=False, resign_threshold=-0.9, resign_check_steps=5,
            learning_rate=0.01, weight_decay=1e-4, lr_milestones=[2], lr_gamma=0.1, max_buffer_size=40, min_buffer_size=4,
            num_epochs=1, batch_size=8, eval_games=4, eval_simulations=4, eval_win_rate=0.55, save_interval=2,
            num_iterations=4, checkpoint_dir=str(tmp_path / name))

    class Loop(train_loop.AlphaZeroLoop):
        def _play_shard(self, n_games):                 # 3 samples per game, content a function of (iteration, index)
            smp = np.zeros(3 * n_games, dtype=SAMPLE_DTYPE)
            smp["game_seq"] = self.iteration; smp["ply"] = np.arange(3 * n_games); smp["z"] = 1
            res = np.zeros(n_games, dtype=RESULT_DTYPE)
            res["winner"] = 1; res["steps"] = 10 + self.iteration
            return (torch.from_numpy(smp.view(np.uint8).reshape(-1, 640).copy()),
                    torch.from_numpy(res.view(np.uint8).reshape(-1, 16).copy()))

        def _arena(self):                               # the candidate is promoted in iteration 2, rejected in iteration 4
            ok = self.iteration == 2
            return {"new_wins": 3 if ok else 1, "old_wins": 1 if ok else 3, "draws": 0, "win_rate": 0.75 if ok else 0.25,
                    "model_updated": ok, "games": None}

    def real_adam_step(model, optimizer, scheduler, buffer, config, generator=None, **kw):
        """a real optimisation step on a loss that depends on the buffer's content and on the generator's order"""
        order = torch.randperm(len(buffer), generator=generator)
        x = torch.zeros((4, 15, 10, 9))
        x.view(4, -1)[:, :4] = order[:16].float().view(4, 4) / 100.0 + float(int(buffer.store[:buffer.count].to(torch.int64).sum()) % 1000) / 1000.0
        model.train()
        logits, value = model(x)
        loss = logits.square().mean() + value.square().mean()
        optimizer.zero_grad(); loss.backward(); optimizer.step(); scheduler.step()
        return {"policy_loss": float(loss.detach()), "value_loss": 0.0, "total_loss": float(loss.detach()), "learning_rate": optimizer.param_groups[0]["lr"]}

    saved = training.train_network
    training.train_network = real_adam_step
    try:
        torch.set_num_threads(1)
        a = Loop(cfg_for("a"), device="cpu", seed=5)
        a.train()                                       # iterations 1..4, uninterrupted
        b1 = Loop(cfg_for("b"), device="cpu", seed=5)
        b1.train(2)                                     # iterations 1..2, then the process "dies"
        b2 = Loop(cfg_for("b"), device="cpu", seed=5)   # a fresh object (same run seed: it keys the batch order and the games)
        with torch.no_grad():                           # ... whose weights are NOT what the checkpoint holds
            for p_ in list(b2.current_model.parameters()) + list(b2.best_model.parameters()):
                p_.add_(1.0)
        info = b2.resume(os.path.join(cfg_for("b").checkpoint_dir, "checkpoint_iter2.pt"))
        assert info["iteration"] == 2 and info["replay_buffer_restored"] and b2.iteration == 2 and b2.total_games == 8
        assert [e["iteration"] for e in b2.training_stats] == [1, 2]
        b2.train()                                      # continues with iteration 3
    finally:
        training.train_network = saved
    assert b2.iteration == a.iteration == 4 and b2.total_games == a.total_games == 16
    for ma, mb in ((a.current_model, b2.current_model), (a.best_model, b2.best_model)):
        for (ka, ta), (kb, tb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert ka == kb and torch.equal(ta, tb), ka
    sa, sb = a.optimizer.state_dict(), b2.optimizer.state_dict()
    assert sa["param_groups"] == sb["param_groups"]
    for k in sa["state"]:
        for name in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(torch.as_tensor(sa["state"][k][name]), torch.as_tensor(sb["state"][k][name]))
    assert a.scheduler.state_dict() == b2.scheduler.state_dict() and a.scheduler.last_epoch == 4
    assert a.optimizer.param_groups[0]["lr"] == pytest.approx(0.001)        # MultiStepLR [2] x 0.1, stepped once per iteration
    assert len(a.buffer) == len(b2.buffer) == 40
    oldest = lambda bf: torch.roll(bf.store, -((bf.head - bf.count) % bf.cap), 0)[:bf.count]
    assert torch.equal(oldest(a.buffer), oldest(b2.buffer))
    strip = lambda st: [{k: (v if k != "time" else 0) for k, v in e.items() if k != "self_play"} for e in st]
    assert strip(a.training_stats) == strip(b2.training_stats) and len(b2.training_stats) == 4
    # a checkpoint WITHOUT the extra replay file (e.g. one written by the reference) resumes like the reference: empty buffer
    os.remove(os.path.join(cfg_for("b").checkpoint_dir, "replay_buffer.pt"))
    b3 = Loop(cfg_for("b"), device="cpu", seed=1)
    info = b3.resume(os.path.join(cfg_for("b").checkpoint_dir, "checkpoint_iter4.pt"))
    assert info["iteration"] == 4 and not info["replay_buffer_restored"] and len(b3.buffer) == 0


def test_parallel_self_play_signature_matches_reference():
    import inspect
    from xiangqi_alphazero_amd import selfplay
    params = list(inspect.signature(selfplay.parallel_self_play).parameters)
    assert params[:5] == ["model", "config", "num_workers", "use_gpu_server", "gpu_device"]   # parallel_selfplay.py:264-270

    class Cfg:
        num_simulations = 8
    with pytest.raises(AttributeError):
        selfplay.parallel_self_play(None, Cfg())


def test_reachable_actions_cover_every_legal_move():
    """The policy head's pruned column set (hip_net.py) must contain every move the rules can generate: all fixture
    positions of the reference, and arbitrary piece placements (pieces on squares the rules never put them on)."""
    from xiangqi_alphazero_amd.sample_format import reachable_actions
    reach = np.zeros(8100, dtype=bool)
    reach[reachable_actions()] = True
    assert reach.sum() == 2550
    for d in (G.corpus(), G.crafted()):
        for i in range(len(d["board"])):
            assert reach[G.moves_of(d, i).astype(np.int64)].all()
    rs = np.random.RandomState(11)
    for _ in range(300):
        b = np.zeros(90, dtype=np.int8)
        sq = rs.choice(90, 34, replace=False)
        b[sq[:32]] = rs.choice([1, 2, 3, 4, 5, 6, 7, -1, -2, -3, -4, -5, -6, -7], 32)
        b[sq[32]], b[sq[33]] = 1, -1                                  # at least one king each
        for side in (1, -1):
            assert reach[np.asarray(O.legal_actions(b, side), dtype=np.int64)].all()


def test_torchscript_export_round_trip_matches_reference_outputs(tmp_path):
    """training/export_model.py:71-85 on a reference-format checkpoint: the traced file, loaded back with torch.jit.load,
    reproduces the REFERENCE network's recorded outputs (nn_golden.npz, 64x3) within 1e-5 and equals the eager module
    bit for bit, at batch 1 (the traced shape) and at the fixture's batch."""
    import torch
    from xiangqi_alphazero_amd import export, model, training, weights
    torch.set_num_threads(2)
    net = model.XiangqiNet(64, 3)
    net.load_state_dict(weights.make_state_dict(64, 3))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[5], gamma=0.1)
    training.save_checkpoint(str(tmp_path), 1, net, net, opt, sch, 0, is_best=True)
    out = export.export_to_torchscript(str(tmp_path / "best_model.pt"), str(tmp_path / "model.ts"))
    ts = torch.jit.load(out)
    g, d = G.nn_golden(), G.corpus()
    states = torch.from_numpy(np.stack([O.encode_state(d["board"][i], int(d["side"][i])) for i in g["corpus_index"]]))
    net.eval()
    with torch.no_grad():
        lt, vt = ts(states)
        le, ve = net(states)
        l1, v1 = ts(states[:1])
    assert lt.shape == (len(states), 8100) and vt.shape == (len(states), 1)
    assert torch.equal(lt, le) and torch.equal(vt, ve)
    np.testing.assert_allclose(l1.numpy(), le[:1].numpy(), atol=2e-6)
    probs = torch.softmax(lt, 1).numpy()
    np.testing.assert_allclose(probs[:, g["sample_idx"]], g["64x3_probs_sample"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(vt.numpy().reshape(-1), g["64x3_value"], rtol=0, atol=1e-5)
    with pytest.raises(RuntimeError):                    # no `onnx` in this image: loud, not silent
        export.export_to_onnx(str(tmp_path / "best_model.pt"), str(tmp_path / "model.onnx"))


def test_bench_launches_its_own_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two fresh ranks itself (the reference fans out its own workers,
    parallel_selfplay.py:337-388).  In this container there is no GPU, so both ranks must refuse to run ("needs a GPU":
    the product path has no CPU fallback) and the launcher must return their non-zero status; a WORLD_SIZE that contradicts
    --gpus is refused before anything else."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2 and r.stdout.strip() == ""
    env["WORLD_SIZE"], env["RANK"] = "4", "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "refusing to report a run of a different size" in r.stderr


def test_bench_workload_label_follows_the_arguments():
    import importlib.util
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("xq_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    mk = lambda g, s, c, n, p=False: types.SimpleNamespace(games=g, sims=s, channels=c, blocks=n, peaked=p)
    assert b.workload_label(mk(8192, 800, 256, 10)).startswith("BASELINE configs[2]")
    assert b.workload_label(mk(1024, 400, 128, 6)).startswith("BASELINE configs[1]")
    assert b.workload_label(mk(8192, 800, 256, 20)).startswith("per-GPU share of BASELINE configs[3]")
    assert b.workload_label(mk(4096, 800, 256, 10)).startswith("custom")
    assert "PEAKED" in b.workload_label(mk(8192, 800, 256, 10, True))
    flops, tower = b.net_flops(256, 10)
    assert abs(flops / 1e6 - 2178.0) < 0.5 and abs(b.net_flops(128, 6)[0] / 1e6 - 369.2) < 0.5


def test_inline_asm_mfmas_keep_their_wait_states():
    """tools/check_asm_mfma_hazards.py: every MFMA issued through inline asm (accumulator tiles pinned to VGPRs) either carries its own
    s_nop or has no vector write of its A/B source registers within the two instructions in front of it in the gfx950 assembly the
    current compiler produces (the compiler pads only the MFMAs it can see)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_asm_mfma_hazards.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "xq_conv.hip" in r.stdout and "xq_train.hip" in r.stdout and "xq_conv_bf16.hip" in r.stdout


def test_use_native_conv_leaves_the_cpu_path_and_the_state_dict_alone():
    """XiangqiNet.use_native_conv(True) only re-routes CUDA tensors: on the CPU the module computes what it computed before (eval and
    train mode; to float32 rounding -- torch picks other CPU convolution kernels for channels-last memory), its state_dict keeps keys, shapes and values, the tower filters stay plain [C, C, 3, 3]; unsupported widths
    are refused; there is no CPU implementation behind native_conv (it raises off the GPU)."""
    import copy
    import torch
    from xiangqi_alphazero_amd import hip, model, native_conv, weights
    net = model.XiangqiNet(64, 2)
    net.load_state_dict(weights.make_state_dict(64, 2, seed=2))
    ref = copy.deepcopy(net)
    net.use_native_conv(True)
    assert all(b.native_conv for b in net.res_blocks) and not any(b.native_conv for b in ref.res_blocks)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
        assert a.shape == b.shape and torch.equal(a, b), k
    assert net.res_blocks[0].conv1.weight.is_contiguous()
    x = (torch.rand(5, 15, 10, 9, generator=torch.Generator().manual_seed(1)) < 0.1).float()
    for mode in (False, True):
        net.train(mode); ref.train(mode)
        (la, va), (lb, vb) = net(x), ref(x)
        assert torch.allclose(la, lb, rtol=0, atol=2e-5) and torch.allclose(va, vb, rtol=0, atol=2e-5)
    with pytest.raises(ValueError):
        model.XiangqiNet(16, 1).use_native_conv(True)
    with pytest.raises(hip.XqError):
        native_conv.conv3x3(torch.zeros(1, 64, 10, 9), torch.zeros(64, 64, 3, 3))
    net.use_native_conv(False)
    assert not any(b.native_conv for b in net.res_blocks)
