"""ctypes binding of the CPU ORACLE (oracle/xq_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libxq_oracle.so")

MAX_MOVES = 200
ACTION_SPACE = 8100
STATE_FLOATS = 15 * 90


def build(force: bool = False) -> str:
    """Compile libxq_oracle.so (and oracle/_ref when /root/reference exists)."""
    src = os.path.join(_HERE, "xq_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "xq_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/training/cython_engine"):
        subprocess.call(["make", "-s", "-C", _HERE, "ref"])
    return _LIB_PATH


class _Game(C.Structure):
    _fields_ = [("board", C.c_int8 * 90), ("player", C.c_int), ("move_count", C.c_int),
                ("no_capture", C.c_int), ("hist_len", C.c_int), ("hist_cap", C.c_int),
                ("hist", C.POINTER(C.c_int8))]


class SearchResult(C.Structure):
    _fields_ = [("n_children", C.c_int), ("actions", C.c_uint16 * MAX_MOVES),
                ("visits", C.c_int32 * MAX_MOVES), ("total_value", C.c_double * MAX_MOVES),
                ("prior", C.c_double * MAX_MOVES), ("prior_is_f64", C.c_int),
                ("root_visits", C.c_int32), ("nodes_created", C.c_int64), ("evals", C.c_int64),
                ("terminal_sims", C.c_int64), ("depth_sum", C.c_int64), ("max_depth", C.c_int32)]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                      C.POINTER(C.c_double))
RANDINT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
CHOICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
DIRICHLET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double))
UNIFORM_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)


class _RandSource(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("randint", RANDINT_FN), ("choice_index", CHOICE_FN),
                ("dirichlet", DIRICHLET_FN), ("uniform", UNIFORM_FN)]


class Config(C.Structure):
    _fields_ = [("num_simulations", C.c_int), ("c_puct", C.c_double),
                ("temperature_threshold", C.c_int), ("max_game_length", C.c_int),
                ("random_opening_moves", C.c_int), ("enable_resign", C.c_int),
                ("resign_threshold", C.c_double), ("resign_check_steps", C.c_int)]


class Sample(C.Structure):
    _fields_ = [("board", C.c_int8 * 90), ("player", C.c_int8), ("z", C.c_int8),
                ("n_moves", C.c_int16), ("temperature", C.c_double),
                ("actions", C.c_uint16 * MAX_MOVES), ("visits", C.c_int32 * MAX_MOVES)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i8p = C.POINTER(C.c_int8)
        L.xqo_find_king.argtypes = [i8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.xqo_is_attacked.argtypes = [i8p, C.c_int, C.c_int, C.c_int]
        L.xqo_is_in_check.argtypes = [i8p, C.c_int]
        L.xqo_generate_legal_moves.argtypes = [i8p, C.c_int, C.POINTER(C.c_uint16)]
        L.xqo_has_legal_moves.argtypes = [i8p, C.c_int]
        L.xqo_material.argtypes = [i8p, C.c_int]
        L.xqo_encode_state.argtypes = [i8p, C.c_int, C.POINTER(C.c_float)]
        L.xqo_encode_state.restype = None
        L.xqo_initial_board.argtypes = [i8p]
        L.xqo_initial_board.restype = None
        L.xqo_game_init.argtypes = [C.POINTER(_Game)]
        L.xqo_game_init.restype = None
        L.xqo_game_free.argtypes = [C.POINTER(_Game)]
        L.xqo_game_free.restype = None
        L.xqo_game_clone.argtypes = [C.POINTER(_Game), C.POINTER(_Game)]
        L.xqo_game_clone.restype = None
        L.xqo_game_make_action.argtypes = [C.POINTER(_Game), C.c_int]
        L.xqo_game_make_action.restype = None
        L.xqo_game_is_over.argtypes = [C.POINTER(_Game), C.POINTER(C.c_int)]
        L.xqo_mcts_search.argtypes = [C.POINTER(_Game), C.c_int, C.c_double, C.POINTER(C.c_double),
                                      EVAL_FN, C.c_void_p, C.POINTER(SearchResult)]
        L.xqo_action_probs.argtypes = [C.POINTER(SearchResult), C.c_double, C.POINTER(C.c_double)]
        L.xqo_action_probs.restype = None
        L.xqo_flip_action.argtypes = [C.c_int]
        L.xqo_choice_from_uniform.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double]
        L.xqo_play_one_game.argtypes = [C.POINTER(Config), EVAL_FN, C.c_void_p, C.POINTER(_RandSource),
                                        C.POINTER(Sample), C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xqo_arena_game.argtypes = [EVAL_FN, C.c_void_p, EVAL_FN, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.xqo_perft.argtypes = [i8p, C.c_int, C.c_int]
        L.xqo_perft.restype = C.c_int64
        _lib = L
    return _lib


def _b(board) -> "C.POINTER(C.c_int8)":
    a = np.ascontiguousarray(board, dtype=np.int8).reshape(90)
    return a, a.ctypes.data_as(C.POINTER(C.c_int8))


# ---- rules ---------------------------------------------------------------------------------

def initial_board() -> np.ndarray:
    a = np.zeros(90, dtype=np.int8)
    lib().xqo_initial_board(a.ctypes.data_as(C.POINTER(C.c_int8)))
    return a.reshape(10, 9)


def find_king(board, player):
    a, p = _b(board)
    kr, kc = C.c_int(), C.c_int()
    ok = lib().xqo_find_king(p, int(player), C.byref(kr), C.byref(kc))
    return (kr.value, kc.value) if ok else None


def is_attacked(board, kr, kc, by_player) -> bool:
    a, p = _b(board)
    return bool(lib().xqo_is_attacked(p, int(kr), int(kc), int(by_player)))


def is_in_check(board, player) -> bool:
    a, p = _b(board)
    return bool(lib().xqo_is_in_check(p, int(player)))


def legal_actions(board, player) -> np.ndarray:
    a, p = _b(board)
    out = np.zeros(MAX_MOVES, dtype=np.uint16)
    n = lib().xqo_generate_legal_moves(p, int(player), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out[:n].copy()


def legal_moves(board, player):
    acts = legal_actions(board, player)
    return [(int(a) // 90 // 9, int(a) // 90 % 9, int(a) % 90 // 9, int(a) % 90 % 9) for a in acts]


def has_legal_moves(board, player) -> bool:
    a, p = _b(board)
    return bool(lib().xqo_has_legal_moves(p, int(player)))


def material(board, player) -> int:
    a, p = _b(board)
    return int(lib().xqo_material(p, int(player)))


def encode_state(board, player) -> np.ndarray:
    a, p = _b(board)
    out = np.empty(STATE_FLOATS, dtype=np.float32)
    lib().xqo_encode_state(p, int(player), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(15, 10, 9)


def perft(board, player, depth) -> int:
    a, p = _b(board)
    return int(lib().xqo_perft(p, int(player), int(depth)))


def flip_action(a: int) -> int:
    return int(lib().xqo_flip_action(int(a)))


def choice_from_uniform(p: np.ndarray, u: float) -> int:
    p = np.ascontiguousarray(p, dtype=np.float64)
    return int(lib().xqo_choice_from_uniform(p.ctypes.data_as(C.POINTER(C.c_double)), p.size, float(u)))


# ---- game ----------------------------------------------------------------------------------

class Game:
    """Mirror of the reference XiangqiGame state (game.py:124-170) held by the C oracle."""

    def __init__(self):
        self._g = _Game()
        lib().xqo_game_init(C.byref(self._g))

    def __del__(self):
        try:
            lib().xqo_game_free(C.byref(self._g))
        except Exception:
            pass

    def clone(self) -> "Game":
        g = Game.__new__(Game)
        g._g = _Game()
        lib().xqo_game_clone(C.byref(g._g), C.byref(self._g))
        return g

    @property
    def board(self) -> np.ndarray:
        return np.frombuffer(self._g.board, dtype=np.int8).reshape(10, 9)

    def set_board(self, board, player=None):
        self.board[:] = np.asarray(board, dtype=np.int8).reshape(10, 9)
        if player is not None:
            self._g.player = int(player)

    current_player = property(lambda s: s._g.player)
    move_count = property(lambda s: s._g.move_count)
    no_capture_count = property(lambda s: s._g.no_capture)
    hist_len = property(lambda s: s._g.hist_len)

    def history(self) -> np.ndarray:
        n = self._g.hist_len
        if n == 0:
            return np.zeros((0, 90), dtype=np.int8)
        return np.ctypeslib.as_array(self._g.hist, shape=(n * 90,)).reshape(n, 90).copy()

    def make_action(self, action: int):
        lib().xqo_game_make_action(C.byref(self._g), int(action))

    def legal_actions(self) -> np.ndarray:
        return legal_actions(self.board, self._g.player)

    def is_game_over(self):
        w = C.c_int()
        done = lib().xqo_game_is_over(C.byref(self._g), C.byref(w))
        return (True, w.value) if done else (False, None)

    def state_for_nn(self) -> np.ndarray:
        return encode_state(self.board, self._g.player)


# ---- evaluator / MCTS -----------------------------------------------------------------------

def make_eval(predict):
    """Wrap a python `predict(state f32[15,10,9]) -> (probs f32[8100], float)` as the C callback."""

    def _cb(_ctx, state_p, probs_p, value_p):
        try:
            state = np.ctypeslib.as_array(state_p, shape=(STATE_FLOATS,)).reshape(15, 10, 9)
            probs, value = predict(state)
            out = np.ctypeslib.as_array(probs_p, shape=(ACTION_SPACE,))
            out[:] = np.asarray(probs, dtype=np.float32)
            value_p[0] = float(value)
            return 0
        except Exception:  # pragma: no cover - surfaced as an error code
            import traceback
            traceback.print_exc()
            return 1

    return EVAL_FN(_cb)


def mcts_search(game: Game, num_simulations: int, predict, c_puct: float = 1.5, noise=None) -> SearchResult:
    res = SearchResult()
    cb = predict if isinstance(predict, EVAL_FN) else make_eval(predict)
    if noise is not None:
        noise = np.ascontiguousarray(noise, dtype=np.float64)
        nptr = noise.ctypes.data_as(C.POINTER(C.c_double))
    else:
        nptr = None
    rc = lib().xqo_mcts_search(C.byref(game._g), int(num_simulations), float(c_puct), nptr, cb, None,
                               C.byref(res))
    if rc != 0:
        raise RuntimeError("oracle evaluator failed")
    return res


def action_probs(res: SearchResult, temperature: float) -> np.ndarray:
    out = np.zeros(ACTION_SPACE, dtype=np.float64)
    lib().xqo_action_probs(C.byref(res), float(temperature), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def play_one_game(cfg: dict, predict, randint, choice_index, dirichlet, uniform, max_samples=512):
    """Run the oracle game loop with injected random draws.  Returns (samples, winner, steps, sims, evals)."""
    c = Config(int(cfg["num_simulations"]), float(cfg["c_puct"]), int(cfg["temperature_threshold"]),
               int(cfg["max_game_length"]), int(cfg["random_opening_moves"]), int(bool(cfg["enable_resign"])),
               float(cfg["resign_threshold"]), int(cfg["resign_check_steps"]))
    cb = predict if isinstance(predict, EVAL_FN) else make_eval(predict)

    def _dir(_ctx, n, out):
        vals = np.asarray(dirichlet(n), dtype=np.float64)
        for i in range(n):
            out[i] = vals[i]

    rs = _RandSource(None, RANDINT_FN(lambda _c, lo, hi: int(randint(lo, hi))),
                     CHOICE_FN(lambda _c, n: int(choice_index(n))), DIRICHLET_FN(_dir),
                     UNIFORM_FN(lambda _c: float(uniform())))
    samples = (Sample * max_samples)()
    winner, steps = C.c_int(), C.c_int()
    sims, evals = C.c_int64(), C.c_int64()
    n = lib().xqo_play_one_game(C.byref(c), cb, None, C.byref(rs), samples, max_samples, C.byref(winner),
                                C.byref(steps), C.byref(sims), C.byref(evals))
    out = []
    for i in range(n):
        s = samples[i]
        m = s.n_moves
        out.append(dict(board=np.frombuffer(s.board, dtype=np.int8).copy(), player=int(s.player), z=int(s.z),
                        temperature=float(s.temperature),
                        actions=np.array(s.actions[:m], dtype=np.uint16),
                        visits=np.array(s.visits[:m], dtype=np.int32)))
    return out, winner.value, steps.value, sims.value, evals.value


def arena_game(predict_new, predict_old, new_is_red: bool, num_simulations: int, max_game_length: int, c_puct: float = 1.5):
    """One evaluation game of the reference's arena (train.py:453-535) -> (winner, steps)."""
    cn = predict_new if isinstance(predict_new, EVAL_FN) else make_eval(predict_new)
    co = predict_old if isinstance(predict_old, EVAL_FN) else make_eval(predict_old)
    w, st = C.c_int(), C.c_int()
    rc = lib().xqo_arena_game(cn, None, co, None, int(bool(new_is_red)), int(num_simulations), float(c_puct),
                              int(max_game_length), C.byref(w), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle arena game failed")
    return w.value, st.value
