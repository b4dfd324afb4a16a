"""B2 evaluator on the hand-written kernels: the residual tower (model.py:99-101; >97 % of the FLOPs at 256x10) runs
through `xq_wino_conv3x3` -- fused Winograd F(2x3,3x3) on the fp32 MFMA with folded-BN bias, ReLU and the skip
connection in its epilogue -- on NHWC activations that ping-pong between preallocated buffers.  The 15->C input
convolution reads the sparse encoder planes directly (`xq_stem_conv`), the heads' 1x1 convolutions are one pass over the
tower output (`xq_heads_1x1`), the value head's two small fully connected layers are `xq_value_head`.  fp32 throughout.

Engine protocol (`evaluate_legal`, what `engine.evaluate_and_expand` calls): the policy head's 2880 -> 8100 layer is
evaluated ONLY at the ordered legal moves of each pending evaluation (`xq_policy_head_legal`, ~40 of 8 100 rows per
position) and handed to `xq_engine_expand_legal` as float32[G, 128]: the dense 32 KB/position logits row of the reference
protocol (model.py:118-124, mcts.py:157-188) is never produced and no library GEMM runs.
Dense protocol (`__call__`): logits over the 2 550 action ids some piece can ever move along (`reachable_actions`, -inf
elsewhere; `engine_policy=True`) or over all 8 100 (`full_policy=True`, what `predict()` -- the reference's
single-position protocol -- always uses); that one linear layer is a ROCm library GEMM through torch.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import hip
from .model import InferenceNet, XiangqiNet
from .sample_format import reachable_actions


class HipResNetEvaluator:
    conv_dtype = "f32"               # HipBf16Evaluator: "bf16" (reduced precision, never the default)

    def __init__(self, net: XiangqiNet, device="cuda", engine_policy: bool = False):
        self.engine_policy = bool(engine_policy)
        if net.num_channels % 64 or 8 % (net.num_channels // 64):
            raise hip.XqError("HipResNetEvaluator: channels must be 64, 128, 256 or 512")
        lib = hip.lib()
        if not hasattr(lib, "xq_wino_conv3x3"):
            raise hip.XqError("libxq_hip.so lacks xq_wino_conv3x3")
        if self.conv_dtype == "bf16" and (net.num_channels % 128 or not hasattr(lib, "xq_wino_conv3x3_bf16")):
            raise hip.XqError("HipBf16Evaluator: channels must be 128, 256 or 512 and libxq_hip.so must export xq_wino_conv3x3_bf16")
        self.device = torch.device(device)
        self.C = net.num_channels
        self.num_res_blocks = net.num_res_blocks
        torch.backends.cuda.matmul.allow_tf32 = False
        torch.backends.cudnn.allow_tf32 = False
        self._bufs = None
        self._logits = None          # engine_policy: persistent [B, 8100] rows, -inf outside the reachable columns
        self._legal = None           # evaluate_legal: persistent [B, 128] legal-move logits
        # XQ_CONV_REVERSE on every other conv launch (Infinity Cache reuse between layers); XQ_CONV_ALTERNATE=0 for A/B runs
        self.alternate_order = os.environ.get("XQ_CONV_ALTERNATE", "1") != "0"
        # conv kernel variant: 128 output channels per workgroup (XQ_CONV_WIDE) where the width allows, else 64
        self.micro_batch = int(os.environ.get("XQ_CONV_MICRO_BATCH", "0"))
        want = os.environ.get("XQ_CONV_BLOCK", "")                 # "64" / "128": force one variant (A/B runs)
        if want in ("64", "128") and self.C % int(want) == 0:
            self.co_blocks = [int(want)]
        else:
            self.co_blocks = [64, 128] if self.C % 128 == 0 else [64]
        if self.conv_dtype == "bf16":
            self.co_blocks = [128]                                  # the bf16 kernel exists in the wide tiling only
        self.reach = torch.from_numpy(reachable_actions()).to(self.device)
        self.timing = False          # bench.py: HIP events around every conv launch of the timed region
        self._events = []
        # bumped whenever a device buffer a recorded step points at is REALLOCATED (a weight tensor whose shape changed, a
        # grown activation / logits buffer): `SelfPlayEngine.step` drops its HIP graph then instead of replaying stale pointers
        self.generation = 0
        self.update(net)

    def update(self, net: XiangqiNet):
        """(Re)build the folded / pre-transformed device weights.  After the first call every tensor is refreshed IN PLACE:
        device pointers stay what they were, so a HIP graph recorded over this evaluator (`engine.capture_step`) keeps
        replaying with the new weights (the reference's `InferenceServer.update_model`, inference_server.py:476-487)."""
        ref = InferenceNet(net)
        new = {}
        new["wt_in"] = hip.stem_weights(ref.w_in)                    # [135, C] for xq_stem_conv
        new["b_in"] = ref.b_in
        for i in range(self.num_res_blocks):
            for cb in self.co_blocks:                                # one pre-transformed copy per kernel variant in use
                new[f"u1_{i}_{cb}"] = self._conv_weights(getattr(ref, f"w1_{i}"), cb)
                new[f"u2_{i}_{cb}"] = self._conv_weights(getattr(ref, f"w2_{i}"), cb)
            new[f"b1_{i}"] = getattr(ref, f"b1_{i}")
            new[f"b2_{i}"] = getattr(ref, f"b2_{i}")
        # both heads' 1x1 convolutions as one [36, C] matrix: rows 0-31 policy, 32-35 value (xq_heads_1x1)
        new["w_pv"] = torch.cat([ref.w_p.view(ref.w_p.shape[0], -1), ref.w_v.view(ref.w_v.shape[0], -1)], 0)
        new["b_pv"] = torch.cat([ref.b_p, ref.b_v], 0)
        # heads consume NHWC rows: permute the FC weights once from the reference's (c, h, w) flatten order to (hw, c)
        fp = ref.fc_p_w.view(-1, 32, 90).permute(0, 2, 1).reshape(-1, 2880)
        fv = ref.fc_v1_w.view(-1, 4, 90).permute(0, 2, 1).reshape(-1, 360)
        new["fc_p_w"], new["fc_p_b"] = fp, ref.fc_p_b
        new["fc_pr_w"], new["fc_pr_b"] = fp[self.reach.cpu()], ref.fc_p_b[self.reach.cpu()]
        new["fc_v1_w"], new["fc_v1_b"] = fv, ref.fc_v1_b
        new["fc_v2_w"], new["fc_v2_b"] = ref.fc_v2_w, ref.fc_v2_b
        new["fc_v1_wt"] = fv.t()                                     # [360, 128] for xq_value_head
        new["fc_v2_vec"] = ref.fc_v2_w.reshape(-1)
        for name, value in new.items():
            value = value.detach().to(self.device, torch.bfloat16 if value.dtype == torch.bfloat16 else torch.float32).contiguous()
            old = getattr(self, name, None)
            if isinstance(old, torch.Tensor) and old.shape == value.shape:
                old.copy_(value)
            else:
                setattr(self, name, value)
                if old is not None:
                    self.generation += 1
        self.blocks_by_variant = {cb: [(getattr(self, f"u1_{i}_{cb}"), getattr(self, f"b1_{i}"), getattr(self, f"u2_{i}_{cb}"),
                                        getattr(self, f"b2_{i}")) for i in range(self.num_res_blocks)] for cb in self.co_blocks}
        self.blocks = self.blocks_by_variant[self.co_blocks[-1]]

    def _conv_weights(self, w: torch.Tensor, co_block: int) -> torch.Tensor:
        return hip.wino_transform_weights(w, co_block)

    def _blocks_for(self, batch: int):
        """Kernel variant by launch size: the wide one (128 output channels per workgroup, one workgroup per CU) pays once a
        launch has at least ~4 rounds of 256 workgroups; smaller launches (BASELINE configs[1]: 1024 games x 128 channels = 480
        wide workgroups) fill the chip better with the narrow one (64 channels, two workgroups per CU)."""
        if 128 in self.blocks_by_variant and 64 in self.blocks_by_variant:
            wide_groups = ((batch * 15 + 31) // 32) * (self.C // 128)
            return self.blocks_by_variant[128 if wide_groups >= 1024 else 64]
        return self.blocks

    def _buffers(self, b):
        """Four NHWC activation buffers, grown to the largest batch seen (callers with a varying batch -- the arena
        evaluates only the searching side's slots -- get prefix views, no reallocation per step)."""
        if self._bufs is None or self._bufs[0].shape[0] < b:
            self.generation += self._bufs is not None
            self._bufs = [torch.empty((b, 90, self.C), dtype=torch.float32, device=self.device) for _ in range(4)]
        return [t[:b] for t in self._bufs]

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, full_policy: bool = False):
        F = torch.nn.functional
        b = x.shape[0]
        p, v = self._tower(x)                                        # stem, residual tower, both heads' 1x1 convolutions
        if self.engine_policy and not full_policy:
            if self._logits is None or self._logits.shape[0] < b:
                self._logits = torch.full((b, hip.ACTION_SPACE), float("-inf"), dtype=torch.float32, device=self.device)
            logits = self._logits[:b]                                # consumed by xq_engine_expand before the next call
            logits[:, self.reach] = F.linear(p.view(b, 2880), self.fc_pr_w, self.fc_pr_b)
        else:
            logits = F.linear(p.view(b, 2880), self.fc_p_w, self.fc_p_b)
        value = hip.value_head(v.view(b, 360), self.fc_v1_wt, self.fc_v1_b, self.fc_v2_vec, self.fc_v2_b)
        return logits, value

    @torch.no_grad()
    def _tower(self, x: torch.Tensor):
        b = x.shape[0]
        x = x.contiguous()
        mb = self.micro_batch
        if mb and b > mb:
            # trial (XQ_CONV_MICRO_BATCH): the tower run over micro-batches whose activations (3 buffers x mb x 90 x C x 4 B) stay
            # inside the 256 MB Infinity Cache from one layer to the next; DESIGN.md section 4.1 has the measurement
            ps, vs = [], []
            for lo in range(0, b, mb):
                p, v = self._tower_once(x[lo:lo + mb])
                ps.append(p.clone()); vs.append(v.clone())
            return torch.cat(ps), torch.cat(vs)
        return self._tower_once(x)

    def _tower_once(self, x: torch.Tensor):
        b = x.shape[0]
        t0, t1, t2, t3 = self._buffers(b)
        h = hip.stem_conv(x, self.wt_in, self.b_in, t0)
        free = [t1, t2, t3]
        rev = self.alternate_order                                   # launches alternate front-to-back / back-to-front:
        for u1, b1, u2, b2 in self._blocks_for(b):                   # each starts on what the previous one wrote last
            y = next(t for t in free if t.data_ptr() != h.data_ptr())
            self._conv(h, u1, b1, y, None, rev)
            o = next(t for t in free if t.data_ptr() != h.data_ptr() and t.data_ptr() != y.data_ptr())
            self._conv(y, u2, b2, o, h, False)
            h = o
        return hip.heads_1x1(h.view(b * 90, self.C), self.w_pv, self.b_pv)

    @torch.no_grad()
    def evaluate_legal(self, x: torch.Tensor, moves: torch.Tensor, counts: torch.Tensor):
        """The engine's protocol (engine.evaluate_and_expand): logits of the ordered legal moves of every pending
        evaluation, float32[G, 128], and the value float32[G] -- every kernel hand-written, no dense policy row.
        Rows of slots that asked for nothing (count 0) are left as they were."""
        b = x.shape[0]
        p, v = self._tower(x)
        if self._legal is None or self._legal.shape[0] < b:
            self.generation += self._legal is not None
            self._legal = torch.zeros((b, hip.MAXM), dtype=torch.float32, device=self.device)
        legal = self._legal[:b]
        hip.policy_head_legal(p.view(b, 2880), self.fc_p_w, self.fc_p_b, moves, counts, legal)
        value = hip.value_head(v.view(b, 360), self.fc_v1_wt, self.fc_v1_b, self.fc_v2_vec, self.fc_v2_b)
        return legal, value

    _conv_launch = staticmethod(hip.wino_conv3x3)

    def _conv(self, x, u, b, out, residual, reverse=False):
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._conv_launch(x, u, b, out, residual, True, reverse)
            e1.record()
            self._events.append((e0, e1))
        else:
            self._conv_launch(x, u, b, out, residual, True, reverse)

    def roofline(self, batch: int, nn_ms: float, launch_ms: float = None):
        """bench.py roofline object for the dominant kernel (k_wino_conv): algorithmic FLOPs of the 3x3 convolution
        it computes (2*90*9*C*C per position, SURVEY.md section 8a row a17 share) x positions per launch, over the
        average launch duration measured with HIP events on the launch stream."""
        if launch_ms is not None:          # no per-launch events (graph replay): the caller's upper bound on a launch
            ms, avg = [], launch_ms
        elif not self._events:
            return None
        else:
            torch.cuda.synchronize(self.device)
            ms = [a.elapsed_time(b) for a, b in self._events]
            avg = sum(ms) / len(ms)
        direct = 2.0 * 90 * 9 * self.C * self.C * batch
        tiles = (batch * 15 + 31) // 32 * 32
        mfma = 20 * 2.0 * tiles * self.C * self.C
        alg = direct / (avg * 1e-3) / 1e12
        issued = mfma / (avg * 1e-3) / 1e12
        # `achieved` / `frac`: the FLOPs the MFMA pipe really issues (Winograd: 300 instead of 810 multiplies per board and
        # channel pair) against the fp32 MFMA peak -- a utilisation, <= 1.  The direct-convolution FLOPs this launch
        # REPLACES (SURVEY.md section 8a row a17's figure) over the same time is `algorithmic_tflops`; its ratio to the
        # peak says how far past a perfect direct implicit GEMM the kernel is, and is not a utilisation.
        peak = self.mfma_peak_tflops
        return {"bound": "mfma", "kernel": self.kernel_label,
                "achieved": round(issued, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(issued / peak, 4), "traffic": None,
                "launches_timed": len(ms), "avg_launch_ms": round(avg, 4),
                "mfma_flops_per_launch": mfma, "algorithmic_flops_per_launch": direct,
                "algorithmic_tflops": round(alg, 2), "algorithmic_speedup_vs_direct": round(alg / peak, 4),
                "share_of_evaluate_ms": round(avg * 2 * self.num_res_blocks / nn_ms, 4)}

    mfma_peak_tflops = 157.3
    kernel_label = "k_wino_conv (fused Winograd F(2x3,3x3) 3x3 conv, fp32 MFMA 32x32x2)"

    def predict(self, state: np.ndarray, device=None):
        x = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.device).unsqueeze(0)
        logits, value = self(x, full_policy=True)
        return torch.softmax(logits, dim=1).squeeze(0).cpu().numpy(), float(value.item())


class HipBf16Evaluator(HipResNetEvaluator):
    """REDUCED-PRECISION throughput mode on a hand-written kernel (hard parts: "keep an fp32 parity mode
    and a bf16 throughput mode, report both"): the residual tower's convolutions run through `xq_wino_conv3x3_bf16`
    (csrc/xq_conv_bf16.hip) -- the same fused Winograd tiling, the input transform computed in float32 and ROUNDED TO bf16,
    bf16 pre-transformed filters, `v_mfma_f32_32x32x16_bf16` with float32 accumulation, float32 bias / skip / ReLU epilogue.
    Activations stay float32 in HBM between layers; the stem, both heads and the engine protocol are the float32 kernels of
    the parent class.  It does NOT meet the 1e-5 contract (bf16 carries 8 significand bits; tests/test_nn_fullsize.py states
    the measured deviation), is never selected by default, and bench.py reports it only as the labelled second object
    `throughput_mode`."""
    conv_dtype = "bf16"
    _conv_launch = staticmethod(hip.wino_conv3x3_bf16)
    mfma_peak_tflops = 2500.0                                        # dense bf16 MFMA peak (MI355X_MICROARCH.md), no sparsity
    kernel_label = "k_wino_conv_bf16 (fused Winograd F(2x3,3x3) 3x3 conv, bf16 MFMA 32x32x16, REDUCED PRECISION)"

    def _conv_weights(self, w: torch.Tensor, co_block: int) -> torch.Tensor:
        return hip.wino_transform_weights_bf16(w)
