"""ctypes loader of libxq_hip.so (C ABI: include/xq_hip.h) plus thin torch-tensor adapters.

The product path has NO CPU fallback: if the library is missing or a call fails, this module raises.
torch is used only for device memory and streams; every computation below runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch  # imported BEFORE the library is loaded: both must share one HIP runtime (libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XQ_HIP_LIB", os.path.join(_HERE, "libxq_hip.so"))   # override: perf experiments only
CSRC = os.path.join(_HERE, "csrc")

MAXM = 128
SAMPLE_BYTES = 640
RESULT_BYTES = 16
ACTION_SPACE = 8100
STATE_FLOATS = 1350


class XqError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cuh", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "xq_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "all"])
    return LIB_PATH


class EngineConfig(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("num_simulations", C.c_int32), ("c_puct", C.c_double),
                ("temperature_threshold", C.c_int32), ("max_game_length", C.c_int32),
                ("random_opening_moves", C.c_int32), ("enable_resign", C.c_int32),
                ("resign_threshold", C.c_double), ("resign_check_steps", C.c_int32), ("add_noise", C.c_int32),
                ("dirichlet_alpha", C.c_double), ("noise_eps", C.c_double), ("late_temperature", C.c_double),
                ("seed", C.c_uint64), ("rank", C.c_int32), ("inject_len", C.c_int32),
                ("games_target", C.c_int64), ("max_out_samples", C.c_int32), ("max_out_results", C.c_int32),
                ("manual_moves", C.c_int32), ("start_stagger", C.c_int32)]


class Engine(C.Structure):
    _fields_ = [("cfg", EngineConfig), ("node_cap", C.c_int32), ("path_cap", C.c_int32),
                ("stage_cap", C.c_int32), ("pad0", C.c_int32), ("p", C.c_void_p * 32)]


class EngineStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("sims", "terminal_sims", "leaf_evals", "root_evals", "moves_played", "games_finished", "red_wins",
                 "black_wins", "draws", "plies_finished", "nodes_created", "depth_sum", "children_scanned", "resigns",
                 "samples_written", "samples_dropped", "overflow", "games_started")] + [("reserved", C.c_uint64 * 14)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


_lib = None


def lib():
    """Load libxq_hip.so; raises XqError when it is absent (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XqError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback for the product path")
    L = C.CDLL(LIB_PATH)
    vp, i32, u64p = C.c_void_p, C.c_int, C.POINTER(C.c_uint64)
    L.xq_version.restype = C.c_char_p
    L.xq_last_hip_error.restype = C.c_char_p
    L.xq_movegen_batch.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.xq_attack_map_batch.argtypes = [vp, i32, vp, vp]
    L.xq_find_king_batch.argtypes = [vp, i32, vp, vp]
    L.xq_encode_batch.argtypes = [vp, vp, i32, vp, vp]
    L.xq_material_batch.argtypes = [vp, i32, vp, vp]
    L.xq_apply_moves_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    L.xq_game_over_batch.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    L.xq_engine_workspace_bytes.argtypes = [C.POINTER(EngineConfig)]
    L.xq_engine_workspace_bytes.restype = C.c_size_t
    L.xq_engine_init.argtypes = [C.POINTER(Engine), C.POINTER(EngineConfig), vp, C.c_size_t, vp, vp]
    L.xq_engine_select.argtypes = [C.POINTER(Engine), vp, vp]
    L.xq_engine_expand.argtypes = [C.POINTER(Engine), vp, vp, i32, vp]
    L.xq_engine_stats_read.argtypes = [C.POINTER(Engine), C.POINTER(EngineStats), vp]
    L.xq_engine_drain.argtypes = [C.POINTER(Engine), vp, i32, C.POINTER(C.c_int), vp, i32, C.POINTER(C.c_int), vp]
    L.xq_engine_drain_device.argtypes = [C.POINTER(Engine), vp, i32, C.POINTER(C.c_int), vp, i32, C.POINTER(C.c_int), vp]
    L.xq_engine_set_position.argtypes = [C.POINTER(Engine), i32, vp, i32, i32, i32, vp, vp, vp]
    L.xq_engine_read_root.argtypes = [C.POINTER(Engine), i32, vp, vp, vp, vp, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    L.xq_bias_act.argtypes = [vp, vp, vp, C.c_longlong, i32, i32, vp]
    L.xq_heads_1x1.argtypes = [vp, vp, vp, vp, vp, C.c_longlong, i32, vp]
    L.xq_stem_conv.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.xq_samples_to_batch.argtypes = [vp, vp, vp, i32, C.c_double, vp, vp, vp, vp]
    L.xq_wino_weight_bytes.argtypes = [i32]
    L.xq_wino_weight_bytes.restype = C.c_size_t
    L.xq_wino_conv3x3.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.xq_wino_transform_filters.argtypes = [vp, vp, i32, i32, vp]
    L.xq_wino_wgrad_scratch_bytes.argtypes = [i32, i32]
    L.xq_wino_wgrad_scratch_bytes.restype = C.c_size_t
    L.xq_wino_wgrad.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.xq_bn_scratch_bytes.argtypes = [i32]
    L.xq_bn_scratch_bytes.restype = C.c_size_t
    L.xq_bn_train_forward.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_longlong, i32, i32, vp, vp, vp, vp, vp, vp]
    L.xq_bn_train_backward.argtypes = [vp, vp, vp, vp, vp, vp, C.c_longlong, i32, i32, vp, vp, vp, vp, vp, vp]
    L.xq_wino_weight_bytes_bf16.argtypes = [i32]
    L.xq_wino_weight_bytes_bf16.restype = C.c_size_t
    L.xq_wino_conv3x3_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.xq_policy_head_legal.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    L.xq_value_head.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    L.xq_engine_requests.argtypes = [C.POINTER(Engine), C.POINTER(vp), C.POINTER(vp)]
    L.xq_engine_expand_legal.argtypes = [C.POINTER(Engine), vp, vp, vp]
    _lib = L
    return L


EXPORTS = ["xq_version", "xq_last_hip_error", "xq_movegen_batch", "xq_attack_map_batch", "xq_find_king_batch",
           "xq_encode_batch", "xq_material_batch", "xq_apply_moves_batch", "xq_game_over_batch",
           "xq_engine_workspace_bytes", "xq_engine_init", "xq_engine_select", "xq_engine_expand",
           "xq_engine_stats_read", "xq_engine_drain", "xq_engine_set_position", "xq_engine_read_root",
           "xq_bias_act", "xq_stem_conv", "xq_heads_1x1", "xq_wino_weight_bytes", "xq_wino_conv3x3", "xq_samples_to_batch",
           "xq_policy_head_legal", "xq_value_head", "xq_engine_requests", "xq_engine_expand_legal", "xq_engine_drain_device",
           "xq_wino_weight_bytes_bf16", "xq_wino_conv3x3_bf16", "xq_wino_transform_filters",
           "xq_bn_scratch_bytes", "xq_bn_train_forward", "xq_bn_train_backward", "xq_wino_wgrad_scratch_bytes", "xq_wino_wgrad"]


def check(rc: int, what: str):
    if rc != 0:
        raise XqError(f"{what} failed: code {rc} ({lib().xq_last_hip_error().decode()})")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda or not t.is_contiguous():
        raise XqError("device-resident contiguous tensor required")
    return t.data_ptr()


# ---- B1 adapters (tensors in, tensors out; all on the current stream) -------------------------------------

def movegen(boards: torch.Tensor, side: torch.Tensor):
    """boards int8[n,90] (or [n,10,9]), side int8[n] -> (moves u16-as-int16 [n,128], counts int16[n],
    in_check uint8[n], status uint8[n])."""
    n = boards.shape[0]
    dev = boards.device
    moves = torch.zeros((n, MAXM), dtype=torch.int16, device=dev)
    counts = torch.zeros(n, dtype=torch.int16, device=dev)
    chk = torch.zeros(n, dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.uint8, device=dev)
    if n:
        check(lib().xq_movegen_batch(_dev(boards), _dev(side), n, _dev(moves), _dev(counts), _dev(chk), _dev(status),
                                     stream_ptr(dev)), "xq_movegen_batch")
    return moves, counts, chk, status


def attack_map(boards: torch.Tensor) -> torch.Tensor:
    n = boards.shape[0]
    out = torch.zeros((n, 2, 90), dtype=torch.uint8, device=boards.device)
    if n:
        check(lib().xq_attack_map_batch(_dev(boards), n, _dev(out), stream_ptr(boards.device)), "xq_attack_map_batch")
    return out


def find_king(boards: torch.Tensor) -> torch.Tensor:
    n = boards.shape[0]
    out = torch.zeros((n, 2), dtype=torch.int16, device=boards.device)
    if n:
        check(lib().xq_find_king_batch(_dev(boards), n, _dev(out), stream_ptr(boards.device)), "xq_find_king_batch")
    return out


def encode(boards: torch.Tensor, side: torch.Tensor) -> torch.Tensor:
    n = boards.shape[0]
    out = torch.empty((n, 15, 10, 9), dtype=torch.float32, device=boards.device)
    if n:
        check(lib().xq_encode_batch(_dev(boards), _dev(side), n, _dev(out), stream_ptr(boards.device)), "xq_encode_batch")
    return out


def material(boards: torch.Tensor) -> torch.Tensor:
    n = boards.shape[0]
    out = torch.zeros((n, 2), dtype=torch.int32, device=boards.device)
    if n:
        check(lib().xq_material_batch(_dev(boards), n, _dev(out), stream_ptr(boards.device)), "xq_material_batch")
    return out


def apply_moves(boards, side, parent, action):
    """parent int32[m] (indices into boards), action int16[m] (u16 ids) -> (child boards int8[m,90], side int8[m])."""
    m = parent.shape[0]
    ob = torch.empty((m, 90), dtype=torch.int8, device=boards.device)
    os_ = torch.empty(m, dtype=torch.int8, device=boards.device)
    if m:
        check(lib().xq_apply_moves_batch(_dev(boards), _dev(side), _dev(parent), _dev(action), m, _dev(ob), _dev(os_),
                                         stream_ptr(boards.device)), "xq_apply_moves_batch")
    return ob, os_


def game_over(boards, side, move_count, no_capture, hist):
    n = boards.shape[0]
    out = torch.zeros((n, 2), dtype=torch.int8, device=boards.device)
    if n:
        check(lib().xq_game_over_batch(_dev(boards), _dev(side), _dev(move_count), _dev(no_capture), _dev(hist), n,
                                       _dev(out), stream_ptr(boards.device)), "xq_game_over_batch")
    return out


def bias_act_(y: torch.Tensor, bias: torch.Tensor, residual=None, relu: bool = True) -> torch.Tensor:
    """In place on a channels-last activation: y = act(y + bias[c] (+ residual)).  `y` is a 4-d tensor whose
    memory is NHWC-contiguous (torch.channels_last) or a 2-d [rows, C] tensor."""
    if y.dim() == 4:
        if not y.is_contiguous(memory_format=torch.channels_last):
            raise XqError("bias_act_: channels_last tensor required")
        c = y.shape[1]
        rows = y.numel() // c
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            raise XqError("bias_act_: channels_last residual required")
    else:
        rows, c = y.shape
        if not y.is_contiguous():
            raise XqError("bias_act_: contiguous tensor required")
    check(lib().xq_bias_act(y.data_ptr(), bias.data_ptr(), None if residual is None else residual.data_ptr(),
                            rows, c, int(relu), stream_ptr(y.device)), "xq_bias_act")
    return y


def stem_weights(w_in: torch.Tensor) -> torch.Tensor:
    """Folded input filters float32[C,15,3,3] -> float32[135, C] (entry = plane*9 + ky*3 + kx), see xq_stem_conv."""
    c = w_in.shape[0]
    if w_in.shape != (c, 15, 3, 3):
        raise XqError("stem_weights: [C,15,3,3] required")
    return w_in.detach().permute(1, 2, 3, 0).reshape(135, c).contiguous()


def stem_conv(planes: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """planes float32[G,15,10,9] contiguous -> out float32[G,90,C] = relu(conv3x3(planes) + bias), NHWC."""
    g = planes.shape[0]
    c = wt.shape[1]
    if planes.shape[1:] != (15, 10, 9) or not planes.is_contiguous() or wt.shape != (135, c) or not wt.is_contiguous() \
            or out.shape != (g, 90, c) or not out.is_contiguous() or planes.dtype != torch.float32:
        raise XqError("stem_conv: planes [G,15,10,9], wt [135,C], out [G,90,C] contiguous float32 required")
    check(lib().xq_stem_conv(planes.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(), g, c,
                             stream_ptr(planes.device)), "xq_stem_conv")
    return out


def heads_1x1(rows: torch.Tensor, w: torch.Tensor, bias: torch.Tensor):
    """rows float32[R, C] (NHWC rows of the tower output), w float32[36, C], bias float32[36] ->
    (policy features float32[R, 32], value features float32[R, 4]), ReLU applied (model.py:43-62)."""
    r, c = rows.shape
    if not rows.is_contiguous() or w.shape != (36, c) or not w.is_contiguous() or bias.shape != (36,):
        raise XqError("heads_1x1: rows [R,C] contiguous, w [36,C], bias [36] required")
    p = torch.empty((r, 32), dtype=torch.float32, device=rows.device)
    v = torch.empty((r, 4), dtype=torch.float32, device=rows.device)
    check(lib().xq_heads_1x1(rows.data_ptr(), w.data_ptr(), bias.data_ptr(), p.data_ptr(), v.data_ptr(), r, c,
                             stream_ptr(rows.device)), "xq_heads_1x1")
    return p, v


def wino_transform_weights(w: torch.Tensor, co_block: int = 64) -> torch.Tensor:
    """Folded 3x3 filters float32[C,C,3,3] -> the kernel's pre-transformed layout (see include/xq_hip.h,
    xq_wino_conv3x3): U = s_p (G_r g G_c'^T)[p][j] per (co, ci) for the F(2,3) x F(3,3) transform, computed in float64,
    stored float32 [C/co_block][C/8][20][2][co_block][4]; co_block = 64 (narrow kernel) or 128 (XQ_CONV_WIDE)."""
    c = w.shape[0]
    if w.shape != (c, c, 3, 3) or co_block not in (64, 128) or c % co_block or 8 % (c // co_block):
        raise XqError("wino_transform_weights: [C,C,3,3] with C a multiple of co_block (64 or 128), C / co_block in {1,2,4,8}")
    gr = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    # F(3,3) at the points 0, 1, -1, 2, inf; the kernel's B^T rows are the textbook ones times (2, 2, 6, 6, 1)
    gc = torch.tensor([[1.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 2.0, 4.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    gc = gc * torch.tensor([0.5, 0.5, 1.0 / 6.0, 1.0 / 6.0, 1.0], dtype=torch.float64)[:, None]
    u = torch.einsum("pr,oirs,qs->pqoi", gr, w.detach().to("cpu", torch.float64), gc)      # [4,5,co,ci]
    u[2] = -u[2]                    # the kernel forms row 2 of B_r^T d as d1 - d2 (the negative of the textbook row)
    u = u.reshape(20, c // co_block, co_block, c // 8, 2, 4)                               # xi, cog, co, chunk, quad, k
    u = u.permute(1, 3, 0, 4, 2, 5).contiguous().to(torch.float32)                         # cog, chunk, xi, quad, co, k
    return u.to(w.device)


def wino_transform_filters_device(w: torch.Tensor, co_block: int = 64, dgrad: bool = False, out: torch.Tensor = None,
                                  both: bool = False) -> torch.Tensor:
    """`wino_transform_weights` by one kernel launch on the device (xq_wino_transform_filters; the train step re-transforms every
    optimizer step).  `dgrad=True`: the filters of the data-gradient convolution, w'[co][ci][r][s] = w[ci][co][2-r][2-s].
    `both=True`: one launch, returns a [2, ...] tensor -- [0] the forward filters, [1] the data-gradient filters."""
    c = w.shape[0]
    if w.shape != (c, c, 3, 3) or w.dtype != torch.float32 or not w.is_cuda or co_block not in (64, 128) or c % co_block \
            or 8 % (c // co_block):
        raise XqError("wino_transform_filters_device: float32[C,C,3,3] on the GPU, C a multiple of co_block (64 or 128), C / co_block in {1,2,4,8}")
    w = w.detach().contiguous()
    shape = (c // co_block, c // 8, 20, 2, co_block, 4)
    if out is None:
        out = torch.empty(((2,) + shape) if both else shape, dtype=torch.float32, device=w.device)
    check(lib().xq_wino_transform_filters(w.data_ptr(), out.data_ptr(), c, (4 if co_block == 128 else 0) | (16 if both else 8 if dgrad else 0),
                                          stream_ptr(w.device)), "xq_wino_transform_filters")
    return out


def wino_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """Weight gradient of the 3x3 convolution (xq_wino_wgrad): x, dy float32[B, 90, C] contiguous -> float32[C, C, 3, 3]."""
    b, n, c = x.shape
    if n != 90 or dy.shape != x.shape or x.dtype != torch.float32 or dy.dtype != torch.float32 or not x.is_contiguous() \
            or not dy.is_contiguous() or not x.is_cuda:
        raise XqError("wino_wgrad: float32[B,90,C] contiguous tensors on the GPU required")
    nbytes = lib().xq_wino_wgrad_scratch_bytes(b, c)
    if nbytes == 0:
        raise XqError("wino_wgrad: channels must be 64, 128, 256 or 512")
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((c, c, 3, 3), dtype=torch.float32, device=x.device)
    check(lib().xq_wino_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), b, c, stream_ptr(x.device)), "xq_wino_wgrad")
    return dw


def wino_transform_weights_bf16(w: torch.Tensor) -> torch.Tensor:
    """Folded 3x3 filters float32[C,C,3,3] -> the bf16 layout of xq_wino_conv3x3_bf16 (REDUCED PRECISION, throughput mode):
    the float32 wide-layout tensor of `wino_transform_weights(w, 128)` regrouped to 16-channel chunks and rounded to bf16,
    bf16[C/128][C/16][20][2][128][8] with input channel 16*chunk + 8*h + k."""
    c = w.shape[0]
    if c % 128 or (c // 16) % 4:
        raise XqError("wino_transform_weights_bf16: C must be 128, 256 or 512")
    u = wino_transform_weights(w, 128)                                 # [cog, chunk8, xi, quad, co, k4]
    ng = c // 128
    u = u.view(ng, c // 16, 2, 20, 2, 128, 4)                          # cog, chunk16, h, xi, quad, co, k4
    u = u.permute(0, 1, 3, 2, 5, 4, 6).contiguous().view(ng, c // 16, 20, 2, 128, 8)   # cog, chunk16, xi, h, co, (quad, k4)
    return u.to(torch.bfloat16).contiguous()


def wino_conv3x3_bf16(x: torch.Tensor, u: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, residual=None,
                      relu: bool = True, reverse: bool = False) -> torch.Tensor:
    """REDUCED-PRECISION convolution (xq_wino_conv3x3_bf16): x, out, residual float32[B, 90, C]; u from
    `wino_transform_weights_bf16`."""
    b, n, c = x.shape
    if n != 90 or not x.is_contiguous() or not out.is_contiguous() or out.shape != x.shape or u.dtype != torch.bfloat16:
        raise XqError("wino_conv3x3_bf16: float32[B,90,C] contiguous tensors and bf16 weights required")
    check(lib().xq_wino_conv3x3_bf16(x.data_ptr(), u.data_ptr(), bias.data_ptr(),
                                     None if residual is None else residual.data_ptr(), out.data_ptr(), b, c,
                                     int(relu) | (2 if reverse else 0), stream_ptr(x.device)), "xq_wino_conv3x3_bf16")
    return out


def wino_conv3x3(x: torch.Tensor, u: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, residual=None,
                 relu: bool = True, reverse: bool = False) -> torch.Tensor:
    """x, out, residual: float32[B, 90, C] contiguous (NHWC); out must not alias x / residual.  `reverse` walks the batch
    back to front (identical results; see XQ_CONV_REVERSE).  The kernel variant follows the weight layout: `u` from
    `wino_transform_weights(w, 128)` (shape [C/128, ...]) selects XQ_CONV_WIDE."""
    b, n, c = x.shape
    if n != 90 or not x.is_contiguous() or not out.is_contiguous() or out.shape != x.shape:
        raise XqError("wino_conv3x3: float32[B,90,C] contiguous tensors required")
    check(lib().xq_wino_conv3x3(x.data_ptr(), u.data_ptr(), bias.data_ptr(),
                                None if residual is None else residual.data_ptr(), out.data_ptr(), b, c,
                                int(relu) | (2 if reverse else 0) | (4 if u.shape[4] == 128 else 0), stream_ptr(x.device)),
          "xq_wino_conv3x3")
    return out



def policy_head_legal(feat: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, moves: torch.Tensor, counts: torch.Tensor,
                      out: torch.Tensor) -> torch.Tensor:
    """feat float32[G, 2880] (NHWC policy features), w float32[8100, 2880] (columns in that order), bias float32[8100],
    moves int16[G, 128] (uint16 action ids), counts int32[G] -> out float32[G, 128]: logits of the listed moves only."""
    g = feat.shape[0]
    if feat.shape != (g, 2880) or w.shape != (ACTION_SPACE, 2880) or moves.shape != (g, MAXM) or counts.shape != (g,) \
            or out.shape != (g, MAXM) or counts.dtype != torch.int32 or moves.element_size() != 2 \
            or not (feat.is_contiguous() and w.is_contiguous() and moves.is_contiguous() and counts.is_contiguous() and out.is_contiguous()):
        raise XqError("policy_head_legal: feat [G,2880], w [8100,2880], moves 16-bit [G,128], counts int32 [G], out [G,128]")
    check(lib().xq_policy_head_legal(feat.data_ptr(), w.data_ptr(), bias.data_ptr(), moves.data_ptr(), counts.data_ptr(), g,
                                     out.data_ptr(), stream_ptr(feat.device)), "xq_policy_head_legal")
    return out


def value_head(vfeat: torch.Tensor, w1t: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """vfeat float32[G, 360] (NHWC value features), w1t float32[360, 128], b1 [128], w2 [128], b2 [1] -> value float32[G]."""
    g = vfeat.shape[0]
    if vfeat.shape != (g, 360) or w1t.shape != (360, 128) or not vfeat.is_contiguous() or not w1t.is_contiguous():
        raise XqError("value_head: vfeat [G,360], w1t [360,128] contiguous required")
    out = torch.empty(g, dtype=torch.float32, device=vfeat.device)
    check(lib().xq_value_head(vfeat.data_ptr(), w1t.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g, out.data_ptr(),
                              stream_ptr(vfeat.device)), "xq_value_head")
    return out
