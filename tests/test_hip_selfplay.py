"""GPU: the B3 operator end to end behind the reference's call shape (parallel_self_play)."""
import numpy as np
import pytest

from oracle import xq_oracle as O

pytestmark = pytest.mark.gpu


class QuickConfig:                       # training/train.py:645-674 ("quick" preset), shortened
    num_simulations = 10
    c_puct = 1.5
    temperature_threshold = 15
    max_game_length = 50
    random_opening_moves = 4
    enable_resign = True
    resign_threshold = -0.85
    resign_check_steps = 3
    num_games_per_iter = 12


def test_parallel_self_play_schema_and_rules():
    import torch
    from xiangqi_alphazero_amd import model, selfplay, weights
    net = model.XiangqiNet(64, 3)
    net.load_state_dict(weights.make_state_dict(64, 3))
    data, stats = selfplay.parallel_self_play(net, QuickConfig(), num_workers=4, use_gpu_server=True,
                                              gpu_device="cuda", n_slots=8, seed=5, return_compact=True)
    for k in ("games", "red_wins", "black_wins", "draws", "avg_steps", "new_samples", "total_time", "num_workers", "mode"):
        assert k in stats                                   # parallel_selfplay.py:316-326
    assert stats["games"] == 12 and stats["mode"] == "hip" and stats["new_samples"] == len(data)
    assert stats["red_wins"] + stats["black_wins"] + stats["draws"] == 12
    smp, res = stats["compact_samples"], stats["compact_results"]
    assert len(data) == 2 * len(smp) and len(res) == 12
    assert sorted(int(r["game_seq"]) for r in res if r["slot"] == 0) == list(range(1, 1 + sum(res["slot"] == 0)))
    for state, pi, z in data[:200]:
        assert state.shape == (15, 10, 9) and state.dtype == np.float32
        assert pi.shape == (8100,) and pi.dtype == np.float64 and abs(pi.sum() - 1.0) < 1e-12
        assert z in (-1.0, 0.0, 1.0)
    # originals: support of pi == legal moves of the sampled position, in any order; mirror: flipped board
    for i in range(0, min(len(smp), 60)):
        s = smp[i]
        n = int(s["n_moves"])
        np.testing.assert_array_equal(s["actions"][:n], O.legal_actions(s["board"], int(s["side"])))
        assert int(s["visits"][:n].sum()) == QuickConfig.num_simulations
    st0, pi0, _ = data[0]
    st1, pi1, _ = data[1]
    np.testing.assert_array_equal(st1, st0[:, :, ::-1])
    assert abs(pi1.sum() - pi0.sum()) < 1e-15 and np.count_nonzero(pi1) == np.count_nonzero(pi0)
    assert stats["avg_steps"] <= 200 and stats["simulations"] >= 12 * QuickConfig.num_simulations
    del torch


def test_graph_replayed_steps_equal_eager_steps():
    """engine.capture_step: a HIP-graph replay of select -> evaluator -> expand must leave the engine in exactly the state
    eager launches do -- same seeds, same network, 300 steps: identical counters, samples and results."""
    import numpy as np
    import torch
    from xiangqi_alphazero_amd import engine, evaluator, model, weights
    net = model.XiangqiNet(64, 3)
    net.load_state_dict(weights.make_state_dict(64, 3, policy_gain=4.0))
    out = []
    for graph in (False, True):
        ev, _ = evaluator.make_evaluator(net, "cuda", "hip")
        cfg = engine.make_config(96, 4, max_game_length=20, random_opening_moves=4, temperature_threshold=8, seed=21)
        eng = engine.SelfPlayEngine(cfg, "cuda", evaluator=ev)
        if graph:
            assert eng.capture_step(warmup=2)
        for _ in range(300 - (2 if graph else 0)):
            eng.step()
        st = eng.stats()
        smp, res = eng.drain()
        out.append((st, smp, res))
    (s0, a0, r0), (s1, a1, r1) = out
    assert s0 == s1 and s0["games_finished"] > 20 and s0["overflow"] == 0
    # the rings are filled in the order games happen to finish inside a launch (an atomic cursor): compare as sets
    a0, a1 = (np.sort(a, order=["slot", "game_seq", "ply"]) for a in (a0, a1))
    r0, r1 = (np.sort(r, order=["slot", "game_seq"]) for r in (r0, r1))
    assert a0.tobytes() == a1.tobytes() and r0.tobytes() == r1.tobytes()
