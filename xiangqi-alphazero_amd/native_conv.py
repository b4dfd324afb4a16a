"""The residual tower's 3x3 convolutions of the TRAIN step on the hand-written Winograd kernel (the reference
trains through torch.nn.Conv2d, training/train.py:376-447 over model.py:20-36).

`WinoConv3x3.apply(x, w)` is `F.conv2d(x, w, None, padding=1)` for float32 [B, C, 10, 9] activations on the GPU:
  * forward: `xq_wino_conv3x3` on the NHWC view of a channels-last tensor (no copy), filters transformed on the device
    (`xq_wino_transform_filters`);
  * data gradient: the same kernel on dL/dy with the transposed, 180-degree-rotated filters (XQ_FILTER_DGRAD);
  * weight gradient: `xq_wino_wgrad` -- the transposed algorithm in the same Winograd domain, summed over tiles on the fp32 MFMA
    (csrc/xq_train.hip).
`BnAct.apply(...)` / `bn_act(x, bn, residual, relu)` is BatchNorm2d in TRAINING mode fused with the ReLU and the skip-add that follow it in a
ResBlock (`xq_bn_train_forward` / `xq_bn_train_backward`, csrc/xq_train.hip): batch statistics and every reduction of the backward pass in
float64 partial sums reduced in a fixed order, running statistics updated in place as torch.nn.BatchNorm2d does.
No CPU fallback: the functions raise off the GPU; `ResBlock` only routes here when `native_conv` is set and the input is a CUDA tensor.
"""
from __future__ import annotations

import os

import torch

from . import hip

_zero_bias = {}


def _co_block(batch: int, channels: int) -> int:
    """Kernel variant by launch size (the rule of HipResNetEvaluator._blocks_for); XQ_TRAIN_CONV_BLOCK=64|128 forces one."""
    want = os.environ.get("XQ_TRAIN_CONV_BLOCK", "")
    if want in ("64", "128") and channels % int(want) == 0:
        return int(want)
    if channels % 128 == 0 and ((batch * 15 + 31) // 32) * (channels // 128) >= 1024:
        return 128
    return 64


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """[B, C, 10, 9] (any strides) -> contiguous [B, 90, C]; free for a channels-last tensor."""
    v = t.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        v = v.contiguous()
    return v.view(t.shape[0], 90, t.shape[1])


def _conv(x_nhwc: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    b, _, c = x_nhwc.shape
    key = (x_nhwc.device, c)
    if key not in _zero_bias:
        _zero_bias[key] = torch.zeros(c, dtype=torch.float32, device=x_nhwc.device)
    out = torch.empty_like(x_nhwc)
    hip.wino_conv3x3(x_nhwc, u, _zero_bias[key], out, None, relu=False)
    return out.view(b, 10, 9, c).permute(0, 3, 1, 2)                      # logical NCHW over channels-last memory


def supported(channels: int) -> bool:
    return channels in (64, 128, 256, 512)


class WinoConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.dtype != torch.float32 or w.dtype != torch.float32 or x.shape[2:] != (10, 9) \
                or w.shape != (x.shape[1], x.shape[1], 3, 3) or not supported(x.shape[1]):
            raise hip.XqError("WinoConv3x3: float32 [B, C, 10, 9] on the GPU with C in {64, 128, 256, 512} and [C, C, 3, 3] filters")
        xv = _nhwc(x)
        need_dx = ctx.needs_input_grad[0]
        # forward and data-gradient filters from ONE transform launch when the input needs a gradient (every tower layer does)
        u = hip.wino_transform_filters_device(w, _co_block(x.shape[0], x.shape[1]), both=need_dx)
        ctx.save_for_backward(x, w, u[1] if need_dx else u)
        return _conv(xv, u[0] if need_dx else u)

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        x, w, u_bwd = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _conv(_nhwc(gy), u_bwd)
        if ctx.needs_input_grad[1]:
            if os.environ.get("XQ_TRAIN_WGRAD", "native") == "library":      # A/B runs only
                gy_cl = gy.contiguous(memory_format=torch.channels_last)
                gw = torch.ops.aten.convolution_backward(gy_cl, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
            else:
                gw = hip.wino_wgrad(_nhwc(x), _nhwc(gy))
        return gx, gw


def conv3x3(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return WinoConv3x3.apply(x, w)


class BnAct(torch.autograd.Function):
    """y = act(batch_norm_train(x) (+ residual)); x, residual, y logical [B, C, 10, 9] over channels-last memory."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, momentum: float, eps: float, relu: bool, batches_tracked=None):
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
            raise hip.XqError("BnAct: float32 [B, C, H, W] on the GPU")
        b, c, h, w = x.shape
        xv = x.permute(0, 2, 3, 1)
        xv = xv if xv.is_contiguous() else xv.contiguous()
        rv = None
        if residual is not None:
            rv = residual.permute(0, 2, 3, 1)
            rv = rv if rv.is_contiguous() else rv.contiguous()
        y = torch.empty_like(xv)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        scratch = torch.empty(hip.lib().xq_bn_scratch_bytes(c) // 8, dtype=torch.float64, device=x.device)
        gamma_c, beta_c = gamma.detach().contiguous(), beta.detach().contiguous()
        hip.check(hip.lib().xq_bn_train_forward(xv.data_ptr(), None if rv is None else rv.data_ptr(), gamma_c.data_ptr(), beta_c.data_ptr(),
                                                None if running_mean is None else running_mean.data_ptr(),
                                                None if running_var is None else running_var.data_ptr(), float(momentum), float(eps),
                                                b * h * w, c, int(relu), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                None if batches_tracked is None else batches_tracked.data_ptr(),
                                                scratch.data_ptr(), hip.stream_ptr(x.device)), "xq_bn_train_forward")
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.save_for_backward(xv, y, gamma_c, mean, invstd)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xv, y, gamma, mean, invstd = ctx.saved_tensors
        b, h, w, c = xv.shape
        gv = gy.permute(0, 2, 3, 1)
        gv = gv if gv.is_contiguous() else gv.contiguous()
        dx = torch.empty_like(xv)
        dres = torch.empty_like(xv) if ctx.has_res else None
        dgamma, dbeta = torch.empty_like(mean), torch.empty_like(mean)
        scratch = torch.empty(hip.lib().xq_bn_scratch_bytes(c) // 8, dtype=torch.float64, device=xv.device)
        hip.check(hip.lib().xq_bn_train_backward(gv.data_ptr(), xv.data_ptr(), y.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                                 invstd.data_ptr(), b * h * w, c, int(ctx.relu), dx.data_ptr(),
                                                 None if dres is None else dres.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                 scratch.data_ptr(), hip.stream_ptr(xv.device)), "xq_bn_train_backward")
        return (dx.permute(0, 3, 1, 2), None if dres is None else dres.permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None, None)


def bn_supported(bn: torch.nn.Module) -> bool:
    """A plain BatchNorm2d in training mode with running statistics and a fixed momentum (SyncBatchNorm -- the DDP step -- and eval mode
    stay with torch)."""
    return type(bn) is torch.nn.BatchNorm2d and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None \
        and bn.num_features in (64, 128, 256, 512, 1024)


def bn_act(x: torch.Tensor, bn: torch.nn.BatchNorm2d, residual=None, relu: bool = True) -> torch.Tensor:
    """`relu(bn(x) + residual)` of a training-mode forward through the fused kernels; `num_batches_tracked` advances as in torch."""
    nbt = bn.num_batches_tracked                                        # incremented inside the statistics kernel (no launch of its own)
    if nbt.dtype != torch.int64 or not nbt.is_cuda:
        raise hip.XqError("bn_act: num_batches_tracked must be an int64 tensor on the GPU")
    return BnAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, nbt)
