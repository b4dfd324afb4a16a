"""Device-resident self-play engine: host-side driver of the B3 entry points (include/xq_hip.h).

One `SelfPlayEngine` owns G concurrent game slots on one GPU.  A step is

    select (HIP)  ->  evaluator over the [G,15,10,9] leaf batch  ->  expand/backup (HIP)

all enqueued on torch's current stream; no host round-trip per simulation and no IPC (the reference's
per-evaluation socket hop, training/inference_server.py:333-349, does not exist here).  torch provides
device memory, the stream and (optionally) the network; search, rules, sampling and bookkeeping are the
hand-written kernels.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np
import torch

from . import hip

from .sample_format import RESULT_DTYPE, SAMPLE_DTYPE, dense_pi  # noqa: E402

assert SAMPLE_DTYPE.itemsize == hip.SAMPLE_BYTES and RESULT_DTYPE.itemsize == hip.RESULT_BYTES


def make_config(n_games: int, num_simulations: int, *, c_puct: float = 1.5, temperature_threshold: int = 20,
                max_game_length: int = 300, random_opening_moves: int = 4, enable_resign: bool = True,
                resign_threshold: float = -0.9, resign_check_steps: int = 5, add_noise: bool = True,
                dirichlet_alpha: float = 0.3, noise_eps: float = 0.25, late_temperature: float = 0.3,
                seed: int = 0, rank: int = 0, inject_len: int = 0, games_target: int = 0,
                max_out_samples: int = 0, max_out_results: int = 0, manual_moves: int = 0,
                start_stagger: bool = False) -> hip.EngineConfig:
    """Defaults are the reference's TrainingConfig (training/train.py:55-111) and hard-coded constants
    (mcts.py:118-121, parallel_selfplay.py:92)."""
    if max_out_samples <= 0:
        max_out_samples = max(4096, 4 * n_games * 64)
    if max_out_results <= 0:
        max_out_results = max(1024, 8 * n_games)
    return hip.EngineConfig(n_games, num_simulations, c_puct, temperature_threshold, max_game_length,
                            random_opening_moves, int(enable_resign), resign_threshold, resign_check_steps,
                            int(add_noise), dirichlet_alpha, noise_eps, late_temperature, seed, rank, inject_len,
                            games_target, max_out_samples, max_out_results, int(manual_moves), int(start_stagger))


class SelfPlayEngine:
    def __init__(self, cfg: hip.EngineConfig, device="cuda", evaluator: Optional[Callable] = None,
                 inject: Optional[np.ndarray] = None):
        if not torch.cuda.is_available():
            raise hip.XqError("SelfPlayEngine needs a GPU: the HIP engine has no CPU fallback")
        self.lib = hip.lib()
        self.device = torch.device(device)
        self.cfg = cfg
        self.G = cfg.n_games
        self.evaluator = evaluator
        nbytes = self.lib.xq_engine_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise hip.XqError("invalid engine configuration")
        self.workspace_bytes = int(nbytes)
        self.ws = torch.empty(self.workspace_bytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.ws.data_ptr() + 255) & ~255
        self._inject = None
        inj_ptr = None
        if cfg.inject_len > 0:
            inj = np.ascontiguousarray(inject, dtype=np.uint64)
            assert inj.shape == (self.G, 4, cfg.inject_len)
            self._inject = torch.from_numpy(inj.view(np.int64)).to(self.device)
            inj_ptr = self._inject.data_ptr()
        self.h = hip.Engine()
        self.nn_input = torch.zeros((self.G, 15, 10, 9), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            hip.check(self.lib.xq_engine_init(C.byref(self.h), C.byref(cfg), base, self.workspace_bytes, inj_ptr,
                                              hip.stream_ptr(self.device)), "xq_engine_init")
        # zero-copy int32 view of the per-slot state words (side to move of the REAL game in column 0, move_count 1,
        # phase 3, simulations done 4): host-side policies such as the arena's model choice read it between stages
        gi_off = int(self.h.p[2]) - int(self.ws.data_ptr())
        self.slot_ints = self.ws[gi_off:gi_off + self.G * 32 * 4].view(torch.int32).view(self.G, 32)
        # sparse hand-off to the evaluator (xq_engine_requests): ordered legal moves + their number per pending evaluation
        pm, pc = C.c_void_p(), C.c_void_p()
        hip.check(self.lib.xq_engine_requests(C.byref(self.h), C.byref(pm), C.byref(pc)), "xq_engine_requests")
        mo, co = int(pm.value) - int(self.ws.data_ptr()), int(pc.value) - int(self.ws.data_ptr())
        self.req_moves = self.ws[mo:mo + self.G * hip.MAXM * 2].view(torch.int16).view(self.G, hip.MAXM)
        self.req_counts = self.ws[co:co + self.G * 4].view(torch.int32)
        self.steps = 0
        self._graph = None
        self._graph_generation = 0
        self.launch_mode = "eager"                     # "graph" once capture_step has recorded a step
        self.capture_error = None

    # ---- the three stages of a step --------------------------------------------------------------------
    def select(self):
        hip.check(self.lib.xq_engine_select(C.byref(self.h), self.nn_input.data_ptr(), hip.stream_ptr(self.device)),
                  "xq_engine_select")
        return self.nn_input

    def expand(self, policy: torch.Tensor, value: torch.Tensor, is_probs: bool = False):
        if policy.dtype != torch.float32 or value.dtype != torch.float32:
            raise hip.XqError("policy/value must be float32")
        policy = policy.contiguous()
        value = value.contiguous().view(-1)
        if policy.shape != (self.G, hip.ACTION_SPACE) or value.shape != (self.G,):
            raise hip.XqError(f"bad evaluator output shapes {tuple(policy.shape)} {tuple(value.shape)}")
        hip.check(self.lib.xq_engine_expand(C.byref(self.h), policy.data_ptr(), value.data_ptr(), int(is_probs),
                                            hip.stream_ptr(self.device)), "xq_engine_expand")
        self._keep = (policy, value)   # keep alive until the stream has consumed them

    def expand_legal(self, legal_logits: torch.Tensor, value: torch.Tensor):
        """Expansion from the logits of the ORDERED LEGAL MOVES only (float32[G, 128], xq_engine_expand_legal)."""
        if legal_logits.dtype != torch.float32 or value.dtype != torch.float32:
            raise hip.XqError("legal_logits/value must be float32")
        legal_logits = legal_logits.contiguous()
        value = value.contiguous().view(-1)
        if legal_logits.shape != (self.G, hip.MAXM) or value.shape != (self.G,):
            raise hip.XqError(f"bad evaluator output shapes {tuple(legal_logits.shape)} {tuple(value.shape)}")
        hip.check(self.lib.xq_engine_expand_legal(C.byref(self.h), legal_logits.data_ptr(), value.data_ptr(),
                                                  hip.stream_ptr(self.device)), "xq_engine_expand_legal")
        self._keep = (legal_logits, value)

    def evaluate_and_expand(self, x: torch.Tensor, evaluator=None):
        """Evaluator -> expansion in the evaluator's own protocol: one that offers `evaluate_legal(x, moves, counts)`
        (the hand-written evaluator) is asked for the legal moves' logits only; any other callable returns the dense
        [G, 8100] logits row of the reference protocol."""
        ev = evaluator if evaluator is not None else self.evaluator
        if hasattr(ev, "evaluate_legal"):
            ll, value = ev.evaluate_legal(x, self.req_moves, self.req_counts)
            self.expand_legal(ll, value)
        else:
            logits, value = ev(x)
            self.expand(logits, value, False)

    def step(self):
        """select -> evaluator -> expand, all asynchronous on the current stream (one graph launch once `capture_step`
        has recorded it)."""
        if self._graph is not None and getattr(self.evaluator, "generation", 0) != self._graph_generation:
            self.release_graph()                       # the evaluator reallocated a buffer: the recording holds stale pointers
        if self._graph is not None:
            self._graph.replay()
        else:
            self.evaluate_and_expand(self.select())
        self.steps += 1

    def capture_step(self, warmup: int = 2) -> bool:
        """Record one step (xq_engine_select, every evaluator kernel, xq_engine_expand[_legal]) into a HIP graph and make
        `step()` replay it: one launch per step instead of ~2B+8 launches issued from Python.  All kernel arguments of a
        step are constants of the engine (the workspace, the evaluator's persistent buffers), so the recording stays valid
        until the evaluator's weights are replaced (`release_graph()` then).  Matters where steps are short: at BASELINE
        configs[1] (1024 games, 128x6) the launches of an eager step leave the GPU idle for ~17 % of it; at configs[2] a
        step is 52 ms and the gain is nil.  `warmup` eager steps run first (allocations, function attributes).  Returns
        False, and stays eager, for evaluators that are not capturable (they synchronise or allocate outside torch)."""
        if self.evaluator is None:
            raise hip.XqError("capture_step needs an evaluator")
        for _ in range(warmup):
            self.evaluate_and_expand(self.select())
            self.steps += 1
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        try:
            # thread_local: other threads of the process (the RCCL watchdog of a multi-rank run polls events) must not
            # invalidate the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.evaluate_and_expand(self.select())
        except hip.XqError:
            raise                                      # argument / shape / launch errors of our own entry points: never hidden
        except RuntimeError as e:
            # ONLY "this evaluator cannot be recorded" (it synchronises, allocates outside torch's pool or launches on
            # another stream while the stream is capturing) keeps the engine eager; anything else is a real error
            msg = str(e).lower()
            if not any(k in msg for k in ("captur", "hiperrorstreamcapture", "cudaerrorstreamcapture", "operation not permitted")):
                raise
            torch.cuda.synchronize(self.device)
            self.launch_mode, self.capture_error = "eager", str(e).splitlines()[0][:200]
            return False
        self._graph = g
        self._graph_generation = getattr(self.evaluator, "generation", 0)
        self.launch_mode = "graph"
        return True

    def release_graph(self):
        self._graph = None
        self.launch_mode = "eager"

    # ---- bookkeeping --------------------------------------------------------------------------------------
    def stats(self, check: bool = True) -> dict:
        s = hip.EngineStats()
        rc = self.lib.xq_engine_stats_read(C.byref(self.h), C.byref(s), hip.stream_ptr(self.device))
        if rc != 0 and (check or rc != -4):
            hip.check(rc, "xq_engine_stats_read")
        return s.as_dict()

    def drain(self):
        """-> (samples structured array, results structured array); empties the device rings."""
        smp = np.zeros(self.cfg.max_out_samples, dtype=SAMPLE_DTYPE)
        res = np.zeros(self.cfg.max_out_results, dtype=RESULT_DTYPE)
        ns, nr = C.c_int(), C.c_int()
        hip.check(self.lib.xq_engine_drain(C.byref(self.h), smp.ctypes.data, len(smp), C.byref(ns), res.ctypes.data,
                                           len(res), C.byref(nr), hip.stream_ptr(self.device)), "xq_engine_drain")
        return smp[:ns.value].copy(), res[:nr.value].copy()

    def drain_device(self):
        """-> (samples uint8[n, 640], results uint8[m, 16]) as DEVICE tensors (xq_engine_drain_device); empties the rings.
        View them with `.cpu().numpy().view(SAMPLE_DTYPE / RESULT_DTYPE)` where host records are wanted."""
        ns, nr = C.c_int(), C.c_int()
        sp = hip.stream_ptr(self.device)
        hip.check(self.lib.xq_engine_drain_device(C.byref(self.h), None, 0, C.byref(ns), None, 0, C.byref(nr), sp),
                  "xq_engine_drain_device")
        smp = torch.empty((ns.value, hip.SAMPLE_BYTES), dtype=torch.uint8, device=self.device)
        res = torch.empty((nr.value, hip.RESULT_BYTES), dtype=torch.uint8, device=self.device)
        if ns.value or nr.value:
            hip.check(self.lib.xq_engine_drain_device(C.byref(self.h), smp.data_ptr() if ns.value else None, ns.value, C.byref(ns),
                                                      res.data_ptr() if nr.value else None, nr.value, C.byref(nr), sp),
                      "xq_engine_drain_device")
        return smp, res

    def arena_views(self) -> dict:
        """Zero-copy torch views of the SoA tree arenas in the workspace, [G, node_cap] each:
        N int32, W float64, P float32, action int16 (uint16 bits), first int32 (first child, -1 = none), meta int16
        (uint16 bits: child count | kind << 14), plus the root boards int8 [G, 96] (90 squares + pad).  For inspection and tests."""
        base = int(self.ws.data_ptr())
        cap = int(self.h.node_cap)

        def view(idx, dtype, cols):
            off = int(self.h.p[idx]) - base
            nbytes = self.G * cols * torch.empty(0, dtype=dtype).element_size()
            return self.ws[off:off + nbytes].view(dtype).view(self.G, cols)

        return dict(N=view(6, torch.int32, cap), W=view(7, torch.float64, cap), P=view(8, torch.float32, cap),
                    action=view(9, torch.int16, cap), first=view(10, torch.int32, cap), meta=view(11, torch.int16, cap),
                    board=view(0, torch.int8, 96), node_cap=cap)

    # ---- MCTS.search for a given position (manual_moves engines; mcts.py:94-155) ---------------------------
    def set_position(self, slot: int, board, side: int, move_count: int = 0, no_capture: int = 0, hist12=None,
                     noise=None):
        b = np.ascontiguousarray(board, dtype=np.int8).reshape(90)
        h = None
        if hist12 is not None and len(hist12):
            h = np.zeros((12, 90), dtype=np.int8)
            hh = np.ascontiguousarray(hist12, dtype=np.int8).reshape(-1, 90)[-12:]
            h[:len(hh)] = hh
        nz = None
        if noise is not None:
            nz = np.zeros(hip.MAXM, dtype=np.float64)
            nz[:len(noise)] = noise
        hip.check(self.lib.xq_engine_set_position(
            C.byref(self.h), slot, b.ctypes.data, int(side), int(move_count), int(no_capture),
            None if h is None else h.ctypes.data, None if nz is None else nz.ctypes.data,
            hip.stream_ptr(self.device)), "xq_engine_set_position")

    def read_root(self, slot: int) -> dict:
        a = np.zeros(hip.MAXM, dtype=np.uint16)
        v = np.zeros(hip.MAXM, dtype=np.int32)
        w = np.zeros(hip.MAXM, dtype=np.float64)
        p = np.zeros(hip.MAXM, dtype=np.float64)
        kind, rv, sd = C.c_int(), C.c_int32(), C.c_int32()
        n = self.lib.xq_engine_read_root(C.byref(self.h), slot, a.ctypes.data, v.ctypes.data, w.ctypes.data,
                                         p.ctypes.data, C.byref(kind), C.byref(rv), C.byref(sd),
                                         hip.stream_ptr(self.device))
        if n < 0:
            hip.check(n, "xq_engine_read_root")
        return dict(actions=a[:n], visits=v[:n], total_value=w[:n], prior=p[:n], prior_is_f64=bool(kind.value),
                    root_visits=rv.value, sims_done=sd.value)


action_probs_dense = dense_pi   # the reference's dense pi (mcts.py:190-206) from compact (action, visit) pairs
