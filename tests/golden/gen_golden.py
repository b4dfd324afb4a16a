#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

    python tests/golden/gen_golden.py [--perft5]

Needs /root/reference (read-only) and oracle/_ref/game_core*.so (`make -C oracle ref`, built by gcc from
the C file the reference ships).  Nothing of the reference is copied: this script imports it, feeds it
seeded inputs and stores inputs + outputs as data.  The GPU box never runs this script.

Fixtures written (all small):
  perft.json            perft(1..4[,5]) from the opening through cy_generate_legal_moves
  corpus.npz            whole seeded playouts, every ply: board, counters, ORDERED legal moves,
                        in-check flags, is_game_over(), material, crc32 of get_state_for_nn(),
                        is_attacked bitmaps for every square and both sides
  crafted.npz           hand-built / synthetic boards (reference known-answer positions of test_v3.py:122-197,
                        the 6-ply line of test_cython.py:62-69, king-less boards, random piece soups)
  mcts_traces.json      reference MCTS.search under tests/stub_eval.py: root visits, W, priors
  game_traces.json      reference _play_one_game with every random draw injected (tests/draws.py)
  flip_perm.npy         8100-entry action permutation of _augment_data
  arena_traces.json     reference AlphaZeroTrainer._serial_evaluate with stub models: per-game winner/steps, totals
  train_trace.json      reference AlphaZeroTrainer.train_network on a recorded game's augmented samples (fixed batch order);
                        train_trace_64x2.json: the same at 64x2 (the smallest net the hand-written convolution takes)
  nn_golden.npz         reference XiangqiNet outputs for generator weights (xiangqi-alphazero_amd/weights.py)
  nn_golden2.npz        the same for 256x20 and for peaked policies (policy_gain 8), plus the reference's priors over
                        the ordered legal moves (MCTS._mask_and_normalize of predict)
"""
import argparse
import importlib.util
import json
import os
import random
import sys
import zlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/training"
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import game as ref_game  # noqa: E402  (reference)
import mcts as ref_mcts  # noqa: E402  (reference)
import game_core as ref_core  # noqa: E402  (reference engine compiled from its own game_core.c)

from draws import Draws  # noqa: E402
from stub_eval import StubEvaluator  # noqa: E402

assert ref_game._USE_CYTHON, "reference must run with its Cython engine"
XiangqiGame = ref_game.XiangqiGame


def _load_weights_mod():
    spec = importlib.util.spec_from_file_location(
        "xq_weights", os.path.join(ROOT, "xiangqi-alphazero_amd", "weights.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ------------------------------------------------------------------------------------------
def perft(board, player, depth):
    moves = ref_core.cy_generate_legal_moves(board, player)
    if depth == 1:
        return len(moves)
    total = 0
    for fr, fc, tr, tc in moves:
        cap = board[tr, tc]
        piece = board[fr, fc]
        board[tr, tc] = piece
        board[fr, fc] = 0
        total += perft(board, -player, depth - 1)
        board[fr, fc] = piece
        board[tr, tc] = cap
    return total


def gen_perft(with5):
    g = XiangqiGame()
    depths = [1, 2, 3, 4] + ([5] if with5 else [])
    out = {str(d): perft(g.board.copy(), 1, d) for d in depths}
    path = os.path.join(HERE, "perft.json")
    if not with5 and os.path.exists(path):
        old = json.load(open(path))
        if "5" in old:
            out["5"] = old["5"]
    json.dump(out, open(path, "w"), indent=1)
    print("perft", out)


# ------------------------------------------------------------------------------------------
def attacked_bitmap(board):
    """[2][90] bools: square attacked by red (index 0) / by black (index 1)."""
    bm = np.zeros((2, 90), dtype=np.uint8)
    for i, by in enumerate((1, -1)):
        for sq in range(90):
            bm[i, sq] = 1 if ref_core.cy_is_attacked(board, sq // 9, sq % 9, by) else 0
    return bm


def record_position(g, rec):
    board = g.board
    rec["board"].append(board.reshape(90).copy())
    rec["side"].append(g.current_player)
    rec["move_count"].append(g.move_count)
    rec["no_capture"].append(g.no_capture_count)
    g._legal_moves_cache = None
    acts = g.get_legal_actions()
    rec["moves"].append(np.array(acts, dtype=np.uint16))
    rec["check_red"].append(1 if g._is_in_check(1) else 0)
    rec["check_black"].append(1 if g._is_in_check(-1) else 0)
    done, winner = g.is_game_over()
    rec["done"].append(1 if done else 0)
    rec["winner"].append(2 if winner is None else winner)
    rec["mat_red"].append(g.get_material_score(1))
    rec["mat_black"].append(g.get_material_score(-1))
    rec["state_crc"].append(zlib.crc32(g.get_state_for_nn().tobytes()) & 0xFFFFFFFF)
    rec["attacked"].append(np.packbits(attacked_bitmap(board), axis=1))
    kr = ref_core.cy_find_king(board, 1)
    kb = ref_core.cy_find_king(board, -1)
    rec["king_red"].append(-1 if kr is None else kr[0] * 9 + kr[1])
    rec["king_black"].append(-1 if kb is None else kb[0] * 9 + kb[1])
    return done, winner, acts


def gen_corpus():
    rec = {k: [] for k in ("game", "ply", "board", "side", "move_count", "no_capture", "moves", "check_red",
                           "check_black", "done", "winner", "mat_red", "mat_black", "state_crc", "attacked",
                           "king_red", "king_black", "taken")}
    kinds = {}
    max_l = 0

    def play(gid, policy, seed, scripted=None, max_plies=400, unchecked=0):
        nonlocal max_l
        rng = random.Random(seed)
        g = XiangqiGame()
        ply = 0
        while True:
            rec["game"].append(gid)
            rec["ply"].append(ply)
            done, winner, acts = record_position(g, rec)
            max_l = max(max_l, len(acts))
            if done or ply >= max_plies:
                rec["taken"].append(65535)
                kind = "none"
                if done:
                    if len(acts) == 0:
                        kind = "no_moves"
                    elif g.no_capture_count >= 120:
                        kind = "no_capture"
                    elif g.move_count >= 200:
                        kind = "ply200_w%d" % winner
                    else:
                        kind = "repetition"
                kinds[kind] = kinds.get(kind, 0) + 1
                return
            if scripted is not None and ply < len(scripted):
                a = scripted[ply]   # make_action() applies anything (game.py:528-550); the reference's
                #                     own test line (test_cython.py:62-69) is not made of legal moves
                assert a in acts or ply < unchecked, (gid, ply, a)
            else:
                caps = [a for a in acts if g.board.reshape(90)[a % 90] != 0]
                quiet = [a for a in acts if g.board.reshape(90)[a % 90] == 0]
                if policy == "capture" and caps and rng.random() < 0.9:
                    a = rng.choice(caps)
                elif policy == "quiet" and quiet:
                    a = rng.choice(quiet)
                else:
                    a = rng.choice(acts)
            rec["taken"].append(a)
            g.make_action(a)
            ply += 1

    gid = 0
    for seed in range(10):
        play(gid, "uniform", 1000 + seed); gid += 1
    for seed in range(14):
        play(gid, "capture", 2000 + seed); gid += 1
    for seed in range(6):
        play(gid, "quiet", 3000 + seed); gid += 1
    # scripted shuffles -> three-fold repetition (game.py:607-614)
    enc = ref_game.encode_action
    a1, a1b = enc(0, 1, 2, 2), enc(2, 2, 0, 1)
    b1, b1b = enc(9, 1, 7, 2), enc(7, 2, 9, 1)
    play(gid, "uniform", 1, scripted=[a1, b1, a1b, b1b] * 4); gid += 1
    # the 6-ply line of the reference's test_cython.py:62-69, then a different shuffle
    line = [enc(2, 1, 4, 2), enc(7, 1, 5, 2), enc(0, 1, 2, 2), enc(9, 1, 7, 2), enc(3, 0, 4, 0), enc(6, 0, 5, 0)]
    c1, c1b = enc(0, 0, 1, 0), enc(1, 0, 0, 0)
    d1, d1b = enc(9, 0, 8, 0), enc(8, 0, 9, 0)
    play(gid, "uniform", 2, scripted=line + [c1, d1, c1b, d1b] * 4, unchecked=6); gid += 1

    n = len(rec["board"])
    offs = np.zeros(n + 1, dtype=np.int64)
    for i, m in enumerate(rec["moves"]):
        offs[i + 1] = offs[i] + len(m)
    np.savez_compressed(
        os.path.join(HERE, "corpus.npz"),
        game=np.array(rec["game"], dtype=np.int32), ply=np.array(rec["ply"], dtype=np.int32),
        board=np.stack(rec["board"]).astype(np.int8), side=np.array(rec["side"], dtype=np.int8),
        move_count=np.array(rec["move_count"], dtype=np.int32),
        no_capture=np.array(rec["no_capture"], dtype=np.int32),
        moves_off=offs, moves_flat=np.concatenate(rec["moves"]).astype(np.uint16),
        check_red=np.array(rec["check_red"], dtype=np.uint8),
        check_black=np.array(rec["check_black"], dtype=np.uint8),
        done=np.array(rec["done"], dtype=np.uint8), winner=np.array(rec["winner"], dtype=np.int8),
        mat_red=np.array(rec["mat_red"], dtype=np.int32), mat_black=np.array(rec["mat_black"], dtype=np.int32),
        state_crc=np.array(rec["state_crc"], dtype=np.uint32),
        attacked=np.stack(rec["attacked"]).astype(np.uint8),
        king_red=np.array(rec["king_red"], dtype=np.int16), king_black=np.array(rec["king_black"], dtype=np.int16),
        taken=np.array(rec["taken"], dtype=np.uint16))
    print("corpus: %d positions, %d games, max legal moves %d, endings %s" % (n, gid, max_l, kinds))


# ------------------------------------------------------------------------------------------
def gen_crafted():
    boards, sides, names = [], [], []

    def add(name, cells, side=1):
        b = np.zeros((10, 9), dtype=np.int8)
        for (r, c), p in cells.items():
            b[r, c] = p
        boards.append(b); sides.append(side); names.append(name)

    # reference known-answer positions, training/test_v3.py:122-197
    add("v3_flying_general", {(0, 4): 1, (9, 4): -1})
    add("v3_rook_check", {(0, 4): 1, (9, 4): -1, (5, 4): -5})
    add("v3_knight_check", {(0, 4): 1, (2, 3): -4})
    add("v3_knight_leg_blocked", {(0, 4): 1, (2, 3): -4, (1, 3): 7})
    add("v3_cannon_check", {(0, 4): 1, (9, 4): -1, (5, 4): 7, (8, 4): -6})
    for n_, c_, s_ in list(zip(names, boards, sides)):
        boards.append(c_.copy()); sides.append(-s_); names.append(n_ + "_black_to_move")
    init = XiangqiGame().board
    for nm, kill in (("no_red_king", [(0, 4)]), ("no_black_king", [(9, 4)]), ("no_kings", [(0, 4), (9, 4)])):
        b = init.copy()
        for rc in kill:
            b[rc] = 0
        for s in (1, -1):
            boards.append(b.copy()); sides.append(s); names.append(nm)
    boards.append(np.zeros((10, 9), dtype=np.int8)); sides.append(1); names.append("empty")

    # synthetic piece soups (kings in palaces, everything else anywhere) -- not reachable, still defined
    rng = random.Random(77)
    pool = [2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 7, 7, 7]
    for i in range(400):
        b = np.zeros((10, 9), dtype=np.int8)
        b[rng.randrange(0, 3), rng.randrange(3, 6)] = 1
        b[rng.randrange(7, 10), rng.randrange(3, 6)] = -1
        for sign in (1, -1):
            for p in rng.sample(pool, rng.randrange(0, len(pool) + 1)):
                for _ in range(20):
                    r, c = rng.randrange(10), rng.randrange(9)
                    if b[r, c] == 0:
                        b[r, c] = sign * p
                        break
        boards.append(b); sides.append(1 if i % 2 == 0 else -1); names.append("soup%d" % i)

    moves, attacked, chk, kings, has = [], [], [], [], []
    for b, s in zip(boards, sides):
        mv = ref_core.cy_generate_legal_moves(b, s)
        moves.append(np.array([ref_game.encode_action(*m) for m in mv], dtype=np.uint16))
        attacked.append(np.packbits(attacked_bitmap(b), axis=1))
        chk.append([1 if ref_core.cy_is_in_check(b, 1) else 0, 1 if ref_core.cy_is_in_check(b, -1) else 0])
        kr, kb = ref_core.cy_find_king(b, 1), ref_core.cy_find_king(b, -1)
        kings.append([-1 if kr is None else kr[0] * 9 + kr[1], -1 if kb is None else kb[0] * 9 + kb[1]])
        has.append(1 if ref_core.cy_has_legal_moves(b, s) else 0)
    offs = np.zeros(len(boards) + 1, dtype=np.int64)
    for i, m in enumerate(moves):
        offs[i + 1] = offs[i] + len(m)
    np.savez_compressed(
        os.path.join(HERE, "crafted.npz"), names=np.array(names), board=np.stack(boards).reshape(-1, 90),
        side=np.array(sides, dtype=np.int8), moves_off=offs,
        moves_flat=(np.concatenate(moves) if offs[-1] else np.zeros(0)).astype(np.uint16),
        attacked=np.stack(attacked).astype(np.uint8), in_check=np.array(chk, dtype=np.uint8),
        kings=np.array(kings, dtype=np.int16), has_moves=np.array(has, dtype=np.uint8))
    print("crafted: %d boards" % len(boards))


# ------------------------------------------------------------------------------------------
def replay(actions):
    g = XiangqiGame()
    for a in actions:
        g.make_action(int(a))
    return g


def gen_mcts_traces():
    corpus = np.load(os.path.join(HERE, "corpus.npz"))
    game, ply, taken, done = corpus["game"], corpus["ply"], corpus["taken"], corpus["done"]

    def line(gid, upto):
        idx = np.where(game == gid)[0]
        return [int(a) for a in taken[idx][:upto]]

    cases = [("opening", [], "flat"), ("opening", [], "peaked")]
    # mid-game positions
    for gid, p in ((0, 30), (3, 61), (12, 40), (25, 90)):
        n = int((game == gid).sum())
        cases.append(("g%d_p%d" % (gid, min(p, n - 3)), line(gid, min(p, n - 3)), "peaked"))
    # positions 2-3 plies before a decided ending: sims reach terminal leaves
    ends = 0
    for gid in np.unique(game):
        idx = np.where(game == gid)[0]
        if done[idx[-1]] and len(idx) > 6:
            back = 2 if ends % 2 == 0 else 3
            cases.append(("g%d_end-%d" % (gid, back), line(gid, len(idx) - 1 - back), "peaked"))
            cases.append(("g%d_end-%d" % (gid, back), line(gid, len(idx) - 1 - back), "flat"))
            ends += 1
        if ends >= 8:
            break
    out = []
    rs = np.random.RandomState(20240611)
    orig_dirichlet = np.random.dirichlet
    for name, actions, shape in cases:
        for sims in (16, 100, 400):
            for noisy in (False, True):
                if noisy and sims == 400:
                    continue
                g = replay(actions)
                if g.is_game_over()[0]:
                    continue
                n_legal = len(g.get_legal_actions())
                eta = rs.dirichlet([0.3] * n_legal) if noisy else None
                if noisy:
                    np.random.dirichlet = lambda alpha, _e=eta: _e.copy()
                try:
                    ev = StubEvaluator(peaked=(shape == "peaked"))
                    m = ref_mcts.MCTS(ev, num_simulations=sims, c_puct=1.5, device="cpu")
                    # search() re-implemented call by call would hide the root; keep the reference's own
                    # search and rebuild the root from a second, identical run through its pieces:
                    root = _search_keep_root(m, g, noisy)
                finally:
                    np.random.dirichlet = orig_dirichlet
                ch = list(root.children.items())
                pri = [c.prior for _, c in ch]
                out.append(dict(
                    name=name, actions=actions, stub=shape, sims=sims, noisy=noisy,
                    eta=None if eta is None else [float(x).hex() for x in eta],
                    root_actions=[int(a) for a, _ in ch], visits=[int(c.visit_count) for _, c in ch],
                    total_value=[float(c.total_value).hex() for _, c in ch],
                    prior=[float(p).hex() for p in pri],
                    prior_type=type(pri[0]).__name__, root_visits=int(root.visit_count), evals=ev.calls,
                    pi_T1=_nz(ref_mcts.MCTS._get_action_probs(root, 1.0)),
                    pi_T03=_nz(ref_mcts.MCTS._get_action_probs(root, 0.3)),
                    pi_T0=_nz(ref_mcts.MCTS._get_action_probs(root, 0))))
    json.dump(out, open(os.path.join(HERE, "mcts_traces.json"), "w"))
    print("mcts traces: %d" % len(out))


def _nz(p):
    idx = np.nonzero(p)[0]
    return dict(idx=[int(i) for i in idx], val=[float(p[i]).hex() for i in idx])


def _search_keep_root(m, game, add_noise):
    """Run the reference MCTS.search() and hand back its root node: MCTSNode is patched for the duration
    of the call so the first node constructed (the root, mcts.py:104) is captured."""
    captured = []
    orig_init = ref_mcts.MCTSNode.__init__

    def spy(self, parent=None, prior=0.0):
        orig_init(self, parent, prior)
        if parent is None and not captured:
            captured.append(self)

    ref_mcts.MCTSNode.__init__ = spy
    try:
        m.search(game, temperature=1.0, add_noise=add_noise)
    finally:
        ref_mcts.MCTSNode.__init__ = orig_init
    return captured[0]


# ------------------------------------------------------------------------------------------
def gen_game_traces():
    import parallel_selfplay as ref_sp  # reference (imports torch)

    class Cfg:
        pass

    setups = [
        ("resign", dict(num_simulations=24, c_puct=1.5, temperature_threshold=6, max_game_length=200,
                        random_opening_moves=4, enable_resign=True, resign_threshold=-0.3,
                        resign_check_steps=3), "flat", 11),
        ("maxlen", dict(num_simulations=20, c_puct=1.5, temperature_threshold=10, max_game_length=30,
                        random_opening_moves=6, enable_resign=False, resign_threshold=-0.9,
                        resign_check_steps=5), "peaked", 12),
        ("natural", dict(num_simulations=16, c_puct=1.5, temperature_threshold=15, max_game_length=300,
                         random_opening_moves=2, enable_resign=False, resign_threshold=-0.9,
                         resign_check_steps=5), "peaked", 13),
        ("resign_late", dict(num_simulations=12, c_puct=1.5, temperature_threshold=4, max_game_length=300,
                             random_opening_moves=8, enable_resign=True, resign_threshold=-0.55,
                             resign_check_steps=2), "peaked", 14),
    ]
    out = []
    for name, cfgd, shape, seed in setups:
        cfg = Cfg()
        for k, v in cfgd.items():
            setattr(cfg, k, v)
        d = Draws(seed)
        saved = (random.randint, random.choice, np.random.dirichlet, np.random.choice)
        random.randint = lambda lo, hi: d.randint(lo, hi)
        random.choice = lambda seq: seq[d.choice_index(len(seq))]
        np.random.dirichlet = lambda alpha: d.dirichlet(len(alpha))

        def choice(n, p=None):
            # numpy RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, side='right')
            cdf = np.asarray(p, dtype=np.float64).cumsum()
            cdf /= cdf[-1]
            return int(cdf.searchsorted(d.uniform(), side="right"))

        np.random.choice = choice
        try:
            ev = StubEvaluator(peaked=(shape == "peaked"))
            data, winner, steps = ref_sp._play_one_game(ev, cfg, "cpu")
            aug = ref_sp._augment_data(data[:2])
        finally:
            random.randint, random.choice, np.random.dirichlet, np.random.choice = saved
        plies = []
        for state, pi, z in data:
            plies.append(dict(state_crc=zlib.crc32(np.ascontiguousarray(state).tobytes()) & 0xFFFFFFFF,
                              pi=_nz(pi), z=float(z)))
        augrec = [dict(state_crc=zlib.crc32(np.ascontiguousarray(s).tobytes()) & 0xFFFFFFFF, pi=_nz(p), z=float(z))
                  for s, p, z in aug]
        out.append(dict(name=name, cfg=cfgd, stub=shape, seed=seed, winner=int(winner), steps=int(steps),
                        evals=ev.calls, plies=plies, augmented_first2=augrec))
        print("game trace %s: winner %d steps %d samples %d evals %d" % (name, winner, steps, len(data), ev.calls))
    json.dump(out, open(os.path.join(HERE, "game_traces.json"), "w"))

    perm = np.zeros(8100, dtype=np.uint16)
    for a in range(8100):
        fr, fc, tr, tc = ref_game.decode_action(a)
        perm[a] = ref_game.encode_action(fr, 8 - fc, tr, 8 - tc)
    np.save(os.path.join(HERE, "flip_perm.npy"), perm)


# ------------------------------------------------------------------------------------------
def gen_nn():
    import torch
    import model as ref_model  # reference

    torch.set_num_threads(4)
    W = _load_weights_mod()
    corpus = np.load(os.path.join(HERE, "corpus.npz"))
    pick = np.linspace(0, len(corpus["board"]) - 1, 12).astype(int)
    states = []
    for i in pick:
        g = XiangqiGame()
        g.board[:] = corpus["board"][i].reshape(10, 9)
        g.current_player = int(corpus["side"][i])
        states.append(g.get_state_for_nn())
    states = np.stack(states)
    sample_idx = (np.arange(256, dtype=np.int64) * 7919 + 13) % 8100
    out = dict(corpus_index=pick, sample_idx=sample_idx)
    for ch, nb in ((64, 3), (128, 6), (256, 10)):
        net = ref_model.XiangqiNet(num_channels=ch, num_res_blocks=nb)
        net.load_state_dict(W.make_state_dict(ch, nb, seed=0))
        net.eval()
        probs, vals = [], []
        for s in states:
            p, v = net.predict(s, "cpu")
            probs.append(p); vals.append(v)
        probs = np.stack(probs)
        with torch.no_grad():
            logits, v2 = net(torch.from_numpy(states))
        tag = "%dx%d" % (ch, nb)
        top = np.argsort(-probs, axis=1)[:, :32]
        out[tag + "_value"] = np.array(vals, dtype=np.float64)
        out[tag + "_probs_sample"] = probs[:, sample_idx].astype(np.float32)
        out[tag + "_top_idx"] = top.astype(np.int32)
        out[tag + "_top_prob"] = np.take_along_axis(probs, top, axis=1).astype(np.float32)
        out[tag + "_logits_sample"] = logits.numpy()[:, sample_idx].astype(np.float32)
        out[tag + "_logits_sum"] = logits.numpy().astype(np.float64).sum(axis=1)
        out[tag + "_batch_value"] = v2.numpy().reshape(-1).astype(np.float32)
        print("nn", tag, "values", np.round(vals[:4], 4), "max prob", probs.max())
    np.savez_compressed(os.path.join(HERE, "nn_golden.npz"), **out)


def gen_nn2():
    """Round-2 NN fixture (nn_golden2.npz): the AlphaZero-scale 256x20 net of BASELINE configs[3], and *peaked* policy
    sets (weights.make_state_dict(policy_gain=8): probabilities O(0.1), so an absolute 1e-5 is a meaningful bound),
    with -- per state -- the reference's priors over the ORDERED legal moves (MCTS._mask_and_normalize of
    net.predict, mcts.py:176-188): what the engine's expansion must reproduce on the real network."""
    import torch
    import model as ref_model  # reference

    torch.set_num_threads(4)
    W = _load_weights_mod()
    corpus = np.load(os.path.join(HERE, "corpus.npz"))
    pick = np.linspace(5, len(corpus["board"]) - 7, 16).astype(int)
    pick = np.array([i for i in pick if corpus["moves_off"][i + 1] > corpus["moves_off"][i]])   # non-terminal boards
    states, legal = [], []
    for i in pick:
        g = XiangqiGame()
        g.board[:] = corpus["board"][i].reshape(10, 9)
        g.current_player = int(corpus["side"][i])
        states.append(g.get_state_for_nn())
        acts = g.get_legal_actions()
        assert list(acts) == list(corpus["moves_flat"][corpus["moves_off"][i]:corpus["moves_off"][i + 1]])
        legal.append(list(acts))
    states = np.stack(states)
    sample_idx = (np.arange(512, dtype=np.int64) * 6151 + 29) % 8100
    maxl = max(len(a) for a in legal)
    out = dict(corpus_index=pick, sample_idx=sample_idx,
               legal_count=np.array([len(a) for a in legal], dtype=np.int32))
    la = np.zeros((len(legal), maxl), dtype=np.int32)
    for i, a in enumerate(legal):
        la[i, :len(a)] = a
    out["legal_actions"] = la
    for ch, nb, gain in ((256, 20, 1.0), (64, 3, 8.0), (128, 6, 8.0), (256, 10, 8.0), (256, 20, 8.0)):
        net = ref_model.XiangqiNet(num_channels=ch, num_res_blocks=nb)
        net.load_state_dict(W.make_state_dict(ch, nb, seed=0, policy_gain=gain))
        net.eval()
        probs, vals, priors = [], [], np.zeros((len(legal), maxl), dtype=np.float32)
        for i, s in enumerate(states):
            p, v = net.predict(s, "cpu")
            probs.append(p); vals.append(v)
            pr = ref_mcts.MCTS._mask_and_normalize(p, legal[i])
            assert list(pr.keys()) == legal[i]
            priors[i, :len(legal[i])] = np.array([pr[a] for a in legal[i]], dtype=np.float32)
        probs = np.stack(probs)
        with torch.no_grad():
            logits, v2 = net(torch.from_numpy(states))
        logits = logits.numpy()
        tag = "%dx%d" % (ch, nb) + ("" if gain == 1.0 else "_pg%d" % int(gain))
        top = np.argsort(-probs, axis=1)[:, :64]
        out[tag + "_value"] = np.array(vals, dtype=np.float64)
        out[tag + "_probs_sample"] = probs[:, sample_idx].astype(np.float32)
        out[tag + "_top_idx"] = top.astype(np.int32)
        out[tag + "_top_prob"] = np.take_along_axis(probs, top, axis=1).astype(np.float32)
        out[tag + "_top_logit"] = np.take_along_axis(logits, top, axis=1).astype(np.float32)
        out[tag + "_logits_sample"] = logits[:, sample_idx].astype(np.float32)
        out[tag + "_logits_legal"] = np.stack([np.pad(logits[i, legal[i]], (0, maxl - len(legal[i]))) for i in range(len(legal))]).astype(np.float32)
        out[tag + "_priors_legal"] = priors
        print("nn2", tag, "values", np.round(vals[:4], 4), "max prob", probs.max(), "median top-1", np.median(probs.max(1)),
              "max prior", priors.max())
    np.savez_compressed(os.path.join(HERE, "nn_golden2.npz"), **out)


def gen_arena():
    """Reference AlphaZeroTrainer._serial_evaluate (train.py:453-535) run unbound on a stand-in `self` whose two models
    are stub evaluators: per-game winner / steps are read from its own log lines, the totals from its return value."""
    import logging
    import re
    import types
    cwd = os.getcwd()
    os.chdir("/tmp")                                   # train.py opens 'training.log' in the cwd at import
    try:
        import train as ref_train                      # reference
    finally:
        os.chdir(cwd)

    class StubModel(StubEvaluator):
        def state_dict(self):
            return {}

        def load_state_dict(self, sd):
            return None

    out = []
    for name, games, sims, max_len, new_peaked in (("a", 6, 24, 60, True), ("b", 4, 40, 24, False), ("c", 4, 16, 200, True)):
        cfg = types.SimpleNamespace(eval_games=games, eval_simulations=sims, c_puct=1.5, max_game_length=max_len,
                                    eval_win_rate=0.55)
        fake = types.SimpleNamespace(config=cfg, device="cpu", current_model=StubModel(peaked=new_peaked),
                                     best_model=StubModel(peaked=not new_peaked))
        lines = []

        class H(logging.Handler):
            def emit(self, record):
                lines.append(record.getMessage())

        h = H()
        ref_train.logger.addHandler(h)
        try:
            stats = ref_train.AlphaZeroTrainer._serial_evaluate(fake)
        finally:
            ref_train.logger.removeHandler(h)
        per_game = []
        for ln in lines:
            m = re.search(r"评估对局 (\d+): .*赢家=(.), 步数=(\d+)", ln)
            if m:
                per_game.append(dict(game=int(m.group(1)) - 1, winner={"红": 1, "黑": -1, "和": 0}[m.group(2)], steps=int(m.group(3))))
        assert len(per_game) == games
        out.append(dict(name=name, eval_games=games, eval_simulations=sims, max_game_length=max_len, new_peaked=new_peaked,
                        stats={k: (bool(v) if isinstance(v, (bool, np.bool_)) else v) for k, v in stats.items()}, games=per_game))
        print("arena", name, stats, [(g["winner"], g["steps"]) for g in per_game])
    json.dump(out, open(os.path.join(HERE, "arena_traces.json"), "w"))


def gen_train():
    """Reference AlphaZeroTrainer.train_network (train.py:376-447) run unbound on a stand-in `self`: replay buffer =
    the reference's own _augment_data of a recorded stub game, 16x1 net with generator weights, DataLoader forced to
    shuffle=False so the batch order is defined.  Records the returned stats and a few weights after the call."""
    import types
    from collections import deque
    import torch
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        import train as ref_train                      # reference
    finally:
        os.chdir(cwd)
    import parallel_selfplay as ref_sp
    import model as ref_model
    torch.set_num_threads(4)
    W = _load_weights_mod()

    class Cfg:
        pass

    t = [x for x in json.load(open(os.path.join(HERE, "game_traces.json"))) if x["name"] == "maxlen"][0]
    cfg = Cfg()
    for k, v in t["cfg"].items():
        setattr(cfg, k, v)
    d = Draws(t["seed"])
    saved = (random.randint, random.choice, np.random.dirichlet, np.random.choice)
    random.randint = lambda lo, hi: d.randint(lo, hi)
    random.choice = lambda seq: seq[d.choice_index(len(seq))]
    np.random.dirichlet = lambda alpha: d.dirichlet(len(alpha))

    def choice(n, p=None):
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return int(cdf.searchsorted(d.uniform(), side="right"))

    np.random.choice = choice
    try:
        data, winner, steps = ref_sp._play_one_game(StubEvaluator(peaked=True), cfg, "cpu")
    finally:
        random.randint, random.choice, np.random.dirichlet, np.random.choice = saved
    aug = ref_sp._augment_data(data)

    # 16x1: the original trace; 64x2: the smallest width the hand-written convolution takes (XiangqiNet.use_native_conv)
    for (ch, nb, fname) in ((16, 1, "train_trace.json"), (64, 2, "train_trace_64x2.json")):
        net = ref_model.XiangqiNet(num_channels=ch, num_res_blocks=nb)
        net.load_state_dict(W.make_state_dict(ch, nb, seed=3))
        tc = types.SimpleNamespace(min_buffer_size=10, num_epochs=2, batch_size=20)
        opt = torch.optim.Adam(net.parameters(), lr=0.002, weight_decay=1e-4)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1, 80], gamma=0.1)
        fake = types.SimpleNamespace(config=tc, replay_buffer=deque(aug, maxlen=50000), current_model=net, optimizer=opt,
                                     scheduler=sch, device="cpu")
        real_loader = ref_train.DataLoader
        ref_train.DataLoader = lambda ds, **kw: real_loader(ds, **{**kw, "shuffle": False})
        try:
            stats = ref_train.AlphaZeroTrainer.train_network(fake)
        finally:
            ref_train.DataLoader = real_loader
        sd = net.state_dict()
        probe = {k: [float(x) for x in sd[k].flatten()[:8].double()] for k in
                 ("input_conv.0.weight", "res_blocks.0.conv2.weight", "policy_head.4.bias", "value_head.6.weight",
                  "input_conv.1.running_mean", "value_head.1.running_var")}
        out = dict(game="maxlen", n_samples=len(aug), net=[ch, nb], seed=3, num_epochs=2, batch_size=20, lr=0.002,
                   weight_decay=1e-4, milestones=[1, 80], gamma=0.1, stats={k: float(v) for k, v in stats.items()},
                   probe=probe, num_batches_tracked=int(sd["input_conv.1.num_batches_tracked"]))
        json.dump(out, open(os.path.join(HERE, fname), "w"))
        print("train", fname, stats)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--perft5", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    steps = dict(perft=lambda: gen_perft(args.perft5), corpus=gen_corpus, crafted=gen_crafted,
                 mcts=gen_mcts_traces, games=gen_game_traces, nn=gen_nn, nn2=gen_nn2, arena=gen_arena, train=gen_train)
    for k, fn in steps.items():
        if not args.only or k in args.only.split(","):
            fn()
