"""Pins the CPU oracle (oracle/xq_oracle.c) against fixtures recorded from the reference itself
(tests/golden/gen_golden.py).  CPU only; the HIP path is compared against the oracle in test_hip_*.py."""
import zlib

import numpy as np
import pytest

import golden_io as G
from draws import Draws
from oracle import xq_oracle as O
from stub_eval import StubEvaluator


def test_perft_matches_reference():
    b = O.initial_board()
    for d, want in G.perft().items():
        if d <= 4:
            assert O.perft(b, 1, d) == want


def test_opening_has_44_moves():            # reference test_v3.py:117-120
    assert len(O.legal_actions(O.initial_board(), 1)) == 44


def _check_static(d, i):
    board, side = d["board"][i], int(d["side"][i])
    np.testing.assert_array_equal(O.legal_actions(board, side), G.moves_of(d, i))
    bits = G.attacked_bits(d, i)
    for k, by in enumerate((1, -1)):
        got = np.array([O.is_attacked(board, sq // 9, sq % 9, by) for sq in range(90)], dtype=np.uint8)
        np.testing.assert_array_equal(got, bits[k])


def test_corpus_ordered_moves_and_attack_maps():
    d = G.corpus()
    for i in range(len(d["board"])):
        _check_static(d, i)


def test_corpus_check_material_kings_state():
    d = G.corpus()
    for i in range(len(d["board"])):
        b, s = d["board"][i], int(d["side"][i])
        assert O.is_in_check(b, 1) == bool(d["check_red"][i])
        assert O.is_in_check(b, -1) == bool(d["check_black"][i])
        assert O.material(b, 1) == d["mat_red"][i] and O.material(b, -1) == d["mat_black"][i]
        kr, kb = O.find_king(b, 1), O.find_king(b, -1)
        assert (-1 if kr is None else kr[0] * 9 + kr[1]) == d["king_red"][i]
        assert (-1 if kb is None else kb[0] * 9 + kb[1]) == d["king_black"][i]
        assert zlib.crc32(O.encode_state(b, s).tobytes()) & 0xFFFFFFFF == d["state_crc"][i]


def test_corpus_replay_counters_and_game_over():
    """make_action / counters / history / is_game_over incl. every terminal kind (game.py:528-616)."""
    d = G.corpus()
    kinds = set()
    for gid in np.unique(d["game"]):
        idx = np.where(d["game"] == gid)[0]
        g = O.Game()
        for i in idx:
            np.testing.assert_array_equal(g.board.reshape(90), d["board"][i])
            assert g.current_player == d["side"][i]
            assert g.move_count == d["move_count"][i] and g.no_capture_count == d["no_capture"][i]
            assert g.hist_len == g.move_count
            done, winner = g.is_game_over()
            assert done == bool(d["done"][i])
            assert (2 if winner is None else winner) == d["winner"][i]
            np.testing.assert_array_equal(g.history()[-12:].reshape(-1, 90) if g.hist_len else
                                          np.zeros((0, 90), np.int8), G.history_tail(d, i))
            if done:
                kinds.add("mate" if len(G.moves_of(d, i)) == 0 else "nocap" if d["no_capture"][i] >= 120
                          else "ply200" if d["move_count"][i] >= 200 else "rep")
            if d["taken"][i] != 65535:
                g.make_action(int(d["taken"][i]))
    assert kinds == {"mate", "nocap", "ply200", "rep"}


def test_crafted_boards():
    d = G.crafted()
    for i in range(len(d["board"])):
        _check_static(d, i)
        b, s = d["board"][i], int(d["side"][i])
        assert [int(O.is_in_check(b, 1)), int(O.is_in_check(b, -1))] == list(d["in_check"][i])
        kr, kb = O.find_king(b, 1), O.find_king(b, -1)
        assert [-1 if kr is None else kr[0] * 9 + kr[1], -1 if kb is None else kb[0] * 9 + kb[1]] == list(d["kings"][i])
        assert O.has_legal_moves(b, s) == bool(d["has_moves"][i])


def test_reference_known_answer_positions():
    """The reference's own hand-written expectations, training/test_v3.py:138-197."""
    d = G.crafted()
    names = list(d["names"])

    def board(name):
        return d["board"][names.index(name)]

    assert O.is_in_check(board("v3_rook_check"), 1) is True
    assert O.is_attacked(board("v3_knight_check"), 0, 4, -1) is True
    assert O.is_attacked(board("v3_knight_leg_blocked"), 0, 4, -1) is False
    assert O.is_attacked(board("v3_cannon_check"), 0, 4, -1) is True


def test_flip_permutation():
    perm = G.flip_perm()
    got = np.array([O.flip_action(a) for a in range(8100)], dtype=np.uint16)
    np.testing.assert_array_equal(got, perm)
    np.testing.assert_array_equal(perm[perm], np.arange(8100))


def _replay(actions):
    g = O.Game()
    for a in actions:
        g.make_action(a)
    return g


@pytest.mark.parametrize("chunk", range(4))
def test_mcts_traces_bit_exact(chunk):
    """Visit counts, W (fp64) and priors equal to the reference MCTS under the stub evaluator."""
    traces = G.mcts_traces()
    for t in traces[chunk::4]:
        g = _replay(t["actions"])
        ev = StubEvaluator(peaked=(t["stub"] == "peaked"))
        noise = None if t["eta"] is None else np.array([G.hexf(x) for x in t["eta"]])
        r = O.mcts_search(g, t["sims"], ev.predict, c_puct=1.5, noise=noise)
        n = r.n_children
        assert list(r.actions[:n]) == t["root_actions"], t["name"]
        assert list(r.visits[:n]) == t["visits"], (t["name"], t["sims"], t["stub"], t["noisy"])
        assert [float(x).hex() for x in r.total_value[:n]] == t["total_value"]
        assert [float(x).hex() for x in r.prior[:n]] == t["prior"]
        assert bool(r.prior_is_f64) == (t["prior_type"] != "float32")
        assert r.root_visits == t["root_visits"] and r.evals == t["evals"]
        for T, key, tol in ((1.0, "pi_T1", 0.0), (0.0, "pi_T0", 0.0), (0.3, "pi_T03", 1e-14)):
            pi = O.action_probs(r, T)
            idx = np.nonzero(pi)[0]
            assert list(idx) == t[key]["idx"]
            want = np.array([G.hexf(x) for x in t[key]["val"]])
            if tol == 0.0:
                np.testing.assert_array_equal(pi[idx], want)
            else:
                np.testing.assert_allclose(pi[idx], want, rtol=tol, atol=0)


def test_mcts_traces_cover_terminal_leaves_and_depth():
    deep = term = 0
    for t in G.mcts_traces():
        if t["sims"] != 100 or t["noisy"]:
            continue
        g = _replay(t["actions"])
        r = O.mcts_search(g, 100, StubEvaluator(peaked=(t["stub"] == "peaked")).predict)
        deep += r.max_depth >= 4
        term += r.terminal_sims > 0
    assert deep >= 3 and term >= 3


def test_game_traces():
    """Whole game loop (parallel_selfplay.py:42-134) with injected draws: samples, z, winner, steps."""
    for t in G.game_traces():
        d = Draws(t["seed"])
        ev = StubEvaluator(peaked=(t["stub"] == "peaked"))
        samples, winner, steps, sims, evals = O.play_one_game(
            t["cfg"], ev.predict, d.randint, d.choice_index, d.dirichlet, d.uniform)
        assert (winner, steps, len(samples)) == (t["winner"], t["steps"], len(t["plies"])), t["name"]
        assert evals == t["evals"]
        for s, want in zip(samples, t["plies"]):
            assert zlib.crc32(O.encode_state(s["board"], s["player"]).tobytes()) & 0xFFFFFFFF == want["state_crc"]
            assert float(s["z"]) == want["z"]
            order = np.argsort(s["actions"], kind="stable")
            acts, vis = s["actions"][order], s["visits"][order].astype(np.float64)
            keep = vis > 0
            assert list(acts[keep]) == want["pi"]["idx"]
            got = vis[keep] ** (1.0 / s["temperature"])
            got = got / got.sum()
            np.testing.assert_allclose(got, [G.hexf(x) for x in want["pi"]["val"]], rtol=1e-14, atol=0)


def test_choice_from_uniform_matches_numpy():
    rs = np.random.RandomState(5)
    for _ in range(200):
        p = rs.dirichlet([0.5] * 40)
        dense = np.zeros(8100)
        dense[rs.choice(8100, 40, replace=False)] = p
        st = rs.get_state()
        u = rs.random_sample()
        rs.set_state(st)
        want = rs.choice(8100, p=dense)
        assert O.choice_from_uniform(dense, u) == want


def test_arena_games_match_reference():
    """Arena gate (train.py:453-535) with stub models: per-game winner and length, totals."""
    for t in G.arena_traces():
        new, old = StubEvaluator(peaked=t["new_peaked"]), StubEvaluator(peaked=not t["new_peaked"])
        new_wins = old_wins = draws = 0
        for g in t["games"]:
            new_is_red = g["game"] % 2 == 0
            w, steps = O.arena_game(new.predict, old.predict, new_is_red, t["eval_simulations"], t["max_game_length"])
            assert (w, steps) == (g["winner"], g["steps"]), (t["name"], g)
            if w == 0:
                draws += 1
            elif (w == 1) == new_is_red:
                new_wins += 1
            else:
                old_wins += 1
        assert (new_wins, old_wins, draws) == (t["stats"]["new_wins"], t["stats"]["old_wins"], t["stats"]["draws"])
