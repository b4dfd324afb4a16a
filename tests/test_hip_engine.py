"""GPU parity of the self-play engine (k_select / k_expand through the C ABI).

* MCTS.search: root visit counts, W (fp64) and priors must equal -- bit for bit -- the fixtures recorded from the
  reference's mcts.py under the stub evaluator, and the CPU oracle on further positions.
* _play_one_game: whole games with every random draw injected must reproduce the reference's recorded games
  (samples, z, winner, steps).
* production RNG: distributional checks (Dirichlet(0.3) moments, visit-count invariants).
"""
import zlib

import numpy as np
import pytest

import golden_io as G
from draws import Draws, Stream
from oracle import xq_oracle as O
from stub_eval import predict_from_key, state_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    import torch
    from xiangqi_alphazero_amd import engine, hip
    hip.lib()
    assert torch.cuda.is_available()
    return engine


def _replay(actions):
    g = O.Game()
    for a in actions:
        g.make_action(a)
    return g


def _stub_batch(nn_input_host, peaked_flags):
    """Host-side stub evaluator over a batch of planes -> (probs f32[G,8100], values f32[G])."""
    n = nn_input_host.shape[0]
    probs = np.empty((n, 8100), dtype=np.float32)
    vals = np.empty(n, dtype=np.float32)
    cache = {}
    for i in range(n):
        key = (state_key(nn_input_host[i]), bool(peaked_flags[i]))
        if key not in cache:
            cache[key] = predict_from_key(*key)
        probs[i], vals[i] = cache[key]
    return probs, vals


def _run_steps(eng, n_steps, peaked_flags, stop=None):
    import torch
    for s in range(n_steps):
        x = eng.select().cpu().numpy()
        p, v = _stub_batch(x, peaked_flags)
        eng.expand(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), is_probs=True)
        if stop is not None and s % 32 == 31 and stop():
            break


def _set_from_game(eng, slot, g, noise=None):
    eng.set_position(slot, g.board, g.current_player, g.move_count, g.no_capture_count, g.history()[-12:], noise)


@pytest.mark.parametrize("sims", [16, 100, 400])
def test_mcts_search_bit_exact_vs_reference_traces(eng_mod, sims):
    traces = [t for t in G.mcts_traces() if t["sims"] == sims]
    cfg = eng_mod.make_config(len(traces), sims, add_noise=False, manual_moves=True)
    eng = eng_mod.SelfPlayEngine(cfg)
    for i, t in enumerate(traces):
        noise = None if t["eta"] is None else np.array([G.hexf(x) for x in t["eta"]])
        _set_from_game(eng, i, _replay(t["actions"]), noise)
    _run_steps(eng, sims + 1, [t["stub"] == "peaked" for t in traces])
    st = eng.stats()
    assert st["overflow"] == 0
    for i, t in enumerate(traces):
        r = eng.read_root(i)
        tag = (t["name"], t["stub"], t["noisy"])
        assert r["sims_done"] == sims, tag
        assert list(r["actions"]) == t["root_actions"], tag
        assert list(r["visits"]) == t["visits"], tag
        assert [float(x).hex() for x in r["total_value"]] == t["total_value"], tag
        assert [float(x).hex() for x in r["prior"]] == t["prior"], tag
        assert r["prior_is_f64"] == (t["prior_type"] != "float32") and r["root_visits"] == t["root_visits"]
        for T, key in ((1.0, "pi_T1"), (0.0, "pi_T0"), (0.3, "pi_T03")):
            pi = eng_mod.action_probs_dense(r["actions"], r["visits"], T)
            idx = np.nonzero(pi)[0]
            assert list(idx) == t[key]["idx"]
            np.testing.assert_allclose(pi[idx], [G.hexf(x) for x in t[key]["val"]], rtol=0 if T != 0.3 else 1e-14, atol=0)


def test_mcts_search_vs_oracle_on_corpus_positions(eng_mod):
    """64 further positions (every 70th of the corpus), 200 simulations, peaked stub: GPU == oracle exactly."""
    d = G.corpus()
    picks = [i for i in range(5, len(d["board"]), 70) if not d["done"][i]][:64]
    sims = 200
    eng = eng_mod.SelfPlayEngine(eng_mod.make_config(len(picks), sims, add_noise=False, manual_moves=True))
    games = []
    for slot, i in enumerate(picks):
        first = i - d["ply"][i]
        g = _replay([int(a) for a in d["taken"][first:i]])
        np.testing.assert_array_equal(g.board.reshape(90), d["board"][i])
        games.append(g)
        _set_from_game(eng, slot, g)
    _run_steps(eng, sims + 1, [True] * len(picks))
    assert eng.stats()["overflow"] == 0
    from stub_eval import StubEvaluator
    depth_total = 0
    for slot, g in enumerate(games):
        want = O.mcts_search(g, sims, StubEvaluator(peaked=True).predict)
        r = eng.read_root(slot)
        n = want.n_children
        assert list(r["actions"]) == list(want.actions[:n])
        assert list(r["visits"]) == list(want.visits[:n]), slot
        np.testing.assert_array_equal(r["total_value"], np.array(want.total_value[:n]))
        depth_total += want.depth_sum
    st = eng.stats()
    assert st["depth_sum"] == depth_total and st["sims"] == sims * len(picks)


def _inject_array(seed, n_slots, length):
    arr = np.zeros((n_slots, 4, length), dtype=np.uint64)
    for kind in range(4):
        s = Stream(seed, kind + 1)
        arr[:, kind, :] = np.array([s.next_u64() for _ in range(length)], dtype=np.uint64)[None, :]
    return arr


@pytest.mark.parametrize("idx", range(4))
def test_play_one_game_vs_reference_trace(eng_mod, idx):
    t = G.game_traces()[idx]
    c = t["cfg"]
    n_slots, inj_len = 2, 16384
    cfg = eng_mod.make_config(n_slots, c["num_simulations"], c_puct=c["c_puct"],
                              temperature_threshold=c["temperature_threshold"], max_game_length=c["max_game_length"],
                              random_opening_moves=c["random_opening_moves"], enable_resign=c["enable_resign"],
                              resign_threshold=c["resign_threshold"], resign_check_steps=c["resign_check_steps"],
                              add_noise=True, inject_len=inj_len, games_target=n_slots)
    eng = eng_mod.SelfPlayEngine(cfg, inject=_inject_array(t["seed"], n_slots, inj_len))
    peaked = [t["stub"] == "peaked"] * n_slots
    _run_steps(eng, 40000, peaked, stop=lambda: eng.stats()["games_finished"] >= n_slots)
    st = eng.stats()
    assert st["overflow"] == 0 and st["games_finished"] == n_slots
    samples, results = eng.drain()
    assert len(results) == n_slots
    for r in results:
        assert (int(r["winner"]), int(r["steps"]), int(r["n_samples"])) == (t["winner"], t["steps"], len(t["plies"]))
    for slot in range(n_slots):
        mine = samples[samples["slot"] == slot]
        mine = mine[np.argsort(mine["ply"], kind="stable")]
        assert len(mine) == len(t["plies"])
        for s, want in zip(mine, t["plies"]):
            assert zlib.crc32(O.encode_state(s["board"], int(s["side"])).tobytes()) & 0xFFFFFFFF == want["state_crc"]
            assert float(s["z"]) == want["z"]
            n = int(s["n_moves"])
            T = 0.3 if s["late_temp"] else 1.0
            pi = eng_mod.action_probs_dense(s["actions"][:n], s["visits"][:n].astype(np.float64), T)
            nz = np.nonzero(pi)[0]
            assert list(nz) == want["pi"]["idx"]
            np.testing.assert_allclose(pi[nz], [G.hexf(x) for x in want["pi"]["val"]], rtol=0 if T == 1.0 else 1e-14, atol=0)
            assert int(s["visits"][:n].sum()) == c["num_simulations"]


def test_dirichlet_noise_moments(eng_mod):
    """Device Dirichlet(0.3) at the root: eta recovered from the stored fp64 priors has the right moments."""
    n = 2048
    eng = eng_mod.SelfPlayEngine(eng_mod.make_config(n, 4, add_noise=True, manual_moves=True, seed=99))
    g = O.Game()
    for s in range(n):
        _set_from_game(eng, s, g)
    _run_steps(eng, 1, [False] * n)
    probs, _ = predict_from_key(state_key(g.state_for_nn()), False)
    legal = g.legal_actions()
    p = probs[legal]
    ssum = np.float32(0)
    for x in p:
        ssum = np.float32(ssum + x)
    base = (np.float32(0.75) * (p / ssum)).astype(np.float64)
    etas = np.stack([(eng.read_root(s)["prior"] - base) / 0.25 for s in range(0, n, 4)])
    assert etas.min() >= -1e-12 and np.allclose(etas.sum(axis=1), 1.0, atol=1e-9)
    k, a = len(legal), 0.3
    a0 = k * a
    mean, var = 1.0 / k, a * (a0 - a) / (a0 * a0 * (a0 + 1))
    assert abs(etas.mean() - mean) < 1e-9
    assert abs(etas.var() / var - 1.0) < 0.08
    # small-alpha signature, against numpy's own Dirichlet(0.3) (20k draws, k=44): median of the largest
    # component 0.176, mean mass of the top three 0.413
    assert abs(np.median(etas.max(axis=1)) - 0.176) < 0.015
    assert abs(np.mean(np.sort(etas, axis=1)[:, -3:].sum(axis=1)) - 0.413) < 0.02
    # different slots draw different noise
    assert np.abs(etas[0] - etas[1]).max() > 1e-3


def test_selfplay_device_rng_invariants(eng_mod):
    """Free-running self-play with the device RNG: finished games obey the rules' invariants."""
    import torch
    n, sims = 128, 12
    cfg = eng_mod.make_config(n, sims, max_game_length=60, random_opening_moves=6, enable_resign=True,
                              resign_threshold=-0.6, resign_check_steps=2, temperature_threshold=8, seed=7)
    eng = eng_mod.SelfPlayEngine(cfg)
    _run_steps(eng, 1200, [i % 2 == 0 for i in range(n)])
    st = eng.stats()
    assert st["overflow"] == 0 and st["games_finished"] > n // 2
    assert st["red_wins"] + st["black_wins"] + st["draws"] == st["games_finished"]
    samples, results = eng.drain()
    assert len(samples) == st["samples_written"] and st["samples_dropped"] == 0
    assert set(np.unique(samples["z"])).issubset({-1, 0, 1})
    by_game = {(int(r["slot"]), int(r["game_seq"])): r for r in results}
    for s in samples[:: max(1, len(samples) // 400)]:
        n_m = int(s["n_moves"])
        assert int(s["visits"][:n_m].sum()) == sims
        np.testing.assert_array_equal(s["actions"][:n_m], O.legal_actions(s["board"], int(s["side"])))
        r = by_game[(int(s["slot"]), int(s["game_seq"]))]
        w = int(r["winner"])
        assert int(s["z"]) == (0 if w == 0 else (1 if w == int(s["side"]) else -1))
        assert int(s["ply"]) < int(r["steps"]) <= 200
    reasons = set(int(r["reason"]) for r in results)
    assert reasons.issubset({1, 2, 3}) and len(reasons) >= 2
    # openings differ between slots (random_opening_moves draws come from per-slot streams)
    first = samples[samples["game_seq"] == 1]
    assert len({bytes(s["board"]) for s in first[first["ply"] == first["ply"].min()]}) >= 1
    assert len({int(s["ply"]) for s in first}) > 3
    del torch


def test_search_with_real_network_matches_cpu_reference_path(eng_mod):
    """End to end within tolerance: GPU engine + hand-written tower evaluator vs the oracle search driven by the
    torch CPU module's `.predict` (the reference's own evaluator protocol), same generator weights, 64x3 net,
    96 simulations, no noise.  Network outputs agree to ~1e-6, so visit counts may differ only where two PUCT scores
    are within that noise: require exact equality on most positions and a small total deviation."""
    import torch
    from xiangqi_alphazero_amd import evaluator, model, weights
    torch.set_num_threads(8)
    net = model.XiangqiNet(64, 3)
    net.load_state_dict(weights.make_state_dict(64, 3, policy_gain=6.0))     # peaked enough for trees to go deep
    net.eval()
    ev, name = evaluator.make_evaluator(net, "cuda", "hip")
    d = G.corpus()
    picks = [i for i in range(11, len(d["board"]), 400) if not d["done"][i]][:10]
    sims = 96
    eng = eng_mod.SelfPlayEngine(eng_mod.make_config(len(picks), sims, add_noise=False, manual_moves=True), evaluator=ev)
    games = []
    for slot, i in enumerate(picks):
        first = i - d["ply"][i]
        g = _replay([int(a) for a in d["taken"][first:i]])
        games.append(g)
        _set_from_game(eng, slot, g)
    for _ in range(sims + 1):
        eng.step()
    exact, dev, deep = 0, 0, 0
    for slot, g in enumerate(games):
        want = O.mcts_search(g, sims, lambda s: net.predict(s, "cpu"))
        r = eng.read_root(slot)
        n = want.n_children
        assert list(r["actions"]) == list(want.actions[:n]) and r["sims_done"] == sims
        diff = int(np.abs(np.array(r["visits"]) - np.array(want.visits[:n])).sum())
        exact += diff == 0
        dev += diff
        deep += want.max_depth >= 3
        np.testing.assert_allclose(r["prior"], np.array(want.prior[:n]), rtol=2e-4, atol=1e-7)
    assert exact >= len(picks) - 2 and dev <= 8, (exact, dev)
    assert deep >= 3


def test_arena_gate_matches_reference(eng_mod):
    """`_serial_evaluate` on the engine (arena mode) vs the reference's recorded games, stub models, all games
    of a set concurrently: per-game winner and number of plies, win/draw totals."""
    import torch
    from xiangqi_alphazero_amd import arena

    def stub(peaked):
        def f(x):
            p, v = _stub_batch(x.cpu().numpy(), [peaked] * x.shape[0])
            return torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()
        return f

    for t in G.arena_traces():
        res = arena.play_arena(stub(t["new_peaked"]), stub(not t["new_peaked"]), t["eval_games"], t["eval_simulations"],
                               t["max_game_length"], policy_is_probs=True)
        assert [int(r["slot"]) for r in res] == list(range(t["eval_games"]))
        for r, g in zip(res, t["games"]):
            assert (int(r["winner"]), int(r["steps"])) == (g["winner"], g["steps"]), (t["name"], g)
            assert int(r["n_samples"]) == 0


def test_arena_sparse_merge_equals_per_model_dense_evaluation(eng_mod):
    """The arena's product path (both hand-written evaluators over the whole slot batch, the other model's requests masked
    to zero, results merged on the device: no host round trip per step) plays the same games as evaluating each model on
    its own slots through the dense protocol (index_select + xq_engine_expand), for two different real networks."""
    from xiangqi_alphazero_amd import arena, evaluator, model, weights
    nets = []
    for seed in (1, 2):
        n = model.XiangqiNet(64, 2)
        n.load_state_dict(weights.make_state_dict(64, 2, seed=seed, policy_gain=4.0))
        nets.append(n)
    en, _ = evaluator.make_evaluator(nets[0], "cuda", "hip")
    eo, _ = evaluator.make_evaluator(nets[1], "cuda", "hip")
    sparse = arena.play_arena(en, eo, 6, 12, 24)
    dense = arena.play_arena(lambda x: en(x), lambda x: eo(x), 6, 12, 24)        # plain callables: the dense protocol
    assert [int(r["slot"]) for r in sparse] == list(range(6))
    assert [(int(r["winner"]), int(r["steps"])) for r in sparse] == [(int(r["winner"]), int(r["steps"])) for r in dense]
    odd_first = arena.play_arena(en, eo, 3, 12, 24, first_game=1)                 # a shard that starts on an odd game
    assert [(int(r["winner"]), int(r["steps"])) for r in odd_first] == [(int(r["winner"]), int(r["steps"])) for r in sparse[1:4]]


def test_mcts_shim_has_the_reference_call_shape(eng_mod):
    """`MCTS(model, num_simulations, c_puct).search(game, T, add_noise)` / `.get_action` (mcts.py:76-174) on the engine,
    with a duck-typed game object carrying the reference's attribute names."""
    import types
    import torch
    from xiangqi_alphazero_amd import mcts as M
    t = [x for x in G.mcts_traces() if x["sims"] == 100 and not x["noisy"] and x["stub"] == "peaked"][3]
    g = _replay(t["actions"])
    game = types.SimpleNamespace(board=g.board.copy(), current_player=g.current_player, move_count=g.move_count,
                                 no_capture_count=g.no_capture_count, history=[bytes(h) for h in g.history()])

    def stub(x):                                   # batched evaluator returning LOGITS whose softmax is the stub's probs
        p, v = _stub_batch(x.cpu().numpy(), [True] * x.shape[0])
        return torch.log(torch.from_numpy(p).cuda()), torch.from_numpy(v).cuda()

    m = M.MCTS(stub, num_simulations=100, c_puct=1.5)
    pi = m.search(game, temperature=1.0, add_noise=False)
    assert pi.shape == (8100,) and pi.dtype == np.float64 and abs(pi.sum() - 1.0) < 1e-12
    want = np.zeros(8100); want[t["root_actions"]] = t["visits"]; want /= want.sum()
    # log/softmax round trip perturbs the priors by ~1e-7: visit counts agree except on PUCT near-ties
    assert np.abs(pi - want).sum() <= 0.04
    a = m.get_action(game, temperature=0)
    assert pi[a] >= pi.max() - 0.02


@pytest.mark.gpu
def test_arena_graph_and_side_stream_change_nothing(monkeypatch):
    """play_arena's sparse step replayed from a HIP graph with the old network on a forked stream (the shipped path) against the same
    step launched eagerly on one stream: the same kernels on the same data, so every game ends with the same winner after the same
    number of plies."""
    import torch
    from xiangqi_alphazero_amd import arena, evaluator, model, weights
    nets = []
    for seed in (1, 2):
        n = model.XiangqiNet(64, 1)
        n.load_state_dict(weights.make_state_dict(64, 1, seed=seed, policy_gain=4.0))
        nets.append(n)
    out = []
    for graph, streams in (("1", "1"), ("0", "0"), ("1", "0")):
        monkeypatch.setenv("XQ_ARENA_GRAPH", graph)
        monkeypatch.setenv("XQ_ARENA_STREAMS", streams)
        en, _ = evaluator.make_evaluator(nets[0], "cuda", "hip")
        eo, _ = evaluator.make_evaluator(nets[1], "cuda", "hip")
        res = arena.play_arena(en, eo, 6, 24, 60, device="cuda")
        out.append([(int(r["winner"]), int(r["steps"])) for r in res])
    assert out[0] == out[1] == out[2], out
    assert len(out[0]) == 6 and all(s > 0 for _, s in out[0])
