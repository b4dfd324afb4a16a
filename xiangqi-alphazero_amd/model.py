"""Policy/value ResNet evaluator -- host-side mirror of the reference `XiangqiNet` (training/model.py:39-124).

Same constructor arguments, same `state_dict()` keys and shapes (so `best_model.pt` /
`checkpoint_iter*.pt` written by the reference's train.py:537-567 load unchanged), same `forward` /
`predict` contracts.  What differs is how it is *run* for self-play: `InferenceNet` folds every
BatchNorm (eval mode, running statistics) into the preceding convolution once per weight update and
evaluates whole leaf batches that the HIP engine wrote straight into device memory.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROWS, COLS, ACTION_SPACE, IN_PLANES = 10, 9, 8100, 15
POLICY_PLANES, VALUE_PLANES, VALUE_HIDDEN = 32, 4, 128


class ResBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN, skip, ReLU  (model.py:20-36)."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)
        # True: on the GPU the two convolutions run on the hand-written Winograd kernel, forward and data gradient
        # (native_conv.WinoConv3x3); set by XiangqiNet.use_native_conv, never part of the state_dict
        self.native_conv = False

    def _conv(self, conv: nn.Conv2d, x):
        if self.native_conv and x.is_cuda and x.dtype == torch.float32:
            from .native_conv import conv3x3
            return conv3x3(x, conv.weight)
        return conv(x)

    def forward(self, x):
        if self.native_conv and x.is_cuda and x.dtype == torch.float32:
            from .native_conv import bn_act, bn_supported
            if bn_supported(self.bn1) and bn_supported(self.bn2):      # training mode: BatchNorm + ReLU (+ skip) fused, hand-written
                y = bn_act(self._conv(self.conv1, x), self.bn1, None, True)
                return bn_act(self._conv(self.conv2, y), self.bn2, x, True)
        y = F.relu(self.bn1(self._conv(self.conv1, x)))
        y = self.bn2(self._conv(self.conv2, y))
        return F.relu(y + x)


class XiangqiNet(nn.Module):
    """Trainable module; key-compatible with the reference (model.py:48-85)."""

    def __init__(self, num_channels: int = 128, num_res_blocks: int = 6):
        super().__init__()
        self.num_channels = num_channels
        self.num_res_blocks = num_res_blocks
        c = num_channels
        self.input_conv = nn.Sequential(nn.Conv2d(IN_PLANES, c, 3, padding=1, bias=False),
                                        nn.BatchNorm2d(c), nn.ReLU())
        self.res_blocks = nn.ModuleList([ResBlock(c) for _ in range(num_res_blocks)])
        self.policy_head = nn.Sequential(nn.Conv2d(c, POLICY_PLANES, 1, bias=False),
                                         nn.BatchNorm2d(POLICY_PLANES), nn.ReLU(), nn.Flatten(),
                                         nn.Linear(POLICY_PLANES * ROWS * COLS, ACTION_SPACE))
        self.value_head = nn.Sequential(nn.Conv2d(c, VALUE_PLANES, 1, bias=False),
                                        nn.BatchNorm2d(VALUE_PLANES), nn.ReLU(), nn.Flatten(),
                                        nn.Linear(VALUE_PLANES * ROWS * COLS, VALUE_HIDDEN), nn.ReLU(),
                                        nn.Linear(VALUE_HIDDEN, 1), nn.Tanh())

    def use_native_conv(self, on: bool = True) -> "XiangqiNet":
        """Route the residual tower's convolutions of training / eval forwards on the GPU through the hand-written Winograd kernel
        (forward + data gradient; native_conv.py) and keep the module in channels-last memory, which is that kernel's layout.
        Results equal torch's convolution to float32 rounding (tests/test_training.py); state_dict keys and shapes do not change."""
        from .native_conv import supported
        if on and not supported(self.num_channels):
            raise ValueError("native convolution needs 64, 128, 256 or 512 channels")
        for blk in self.res_blocks:
            blk.native_conv = bool(on)
        if on:
            self.to(memory_format=torch.channels_last)
            # ... except the tower's 3x3 filters: the hand-written kernels read and write them as plain [C, C, 3, 3] (a channels-last
            # parameter would be copied to that layout in every filter transform, and its gradient copied back)
            with torch.no_grad():
                for blk in self.res_blocks:
                    for conv in (blk.conv1, blk.conv2):
                        conv.weight.data = conv.weight.data.contiguous()
                        if conv.weight.grad is not None:
                            conv.weight.grad = conv.weight.grad.contiguous()
        return self

    def forward(self, x):
        """x f32[B,15,10,9] -> (policy logits f32[B,8100], value f32[B,1])  (model.py:87-107)."""
        h = None
        if self.res_blocks and self.res_blocks[0].native_conv and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
            from .native_conv import bn_act, bn_supported
            if bn_supported(self.input_conv[1]):
                h = bn_act(self.input_conv[0](x), self.input_conv[1], None, True)
        if h is None:
            h = self.input_conv(x)
        for blk in self.res_blocks:
            h = blk(h)
        return self.policy_head(h), self.value_head(h)

    def predict(self, state: np.ndarray, device: str = "cpu"):
        """Evaluator-plugin protocol (model.py:109-124): one state -> (softmax probs f32[8100], float)."""
        self.eval()
        with torch.no_grad():
            x = torch.as_tensor(np.asarray(state), dtype=torch.float32).unsqueeze(0).to(device)
            logits, value = self(x)
            probs = F.softmax(logits, dim=1).squeeze(0).cpu().numpy()
        return probs, value.item()


def _fold(conv_w: torch.Tensor, bn: nn.BatchNorm2d):
    """BN(eval) o conv  ->  conv with per-output-channel scale and bias."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    bias = bn.bias.detach() - bn.running_mean.detach() * scale
    return conv_w.detach() * scale.view(-1, 1, 1, 1), bias


class InferenceNet(nn.Module):
    """Inference-only restatement with BatchNorm folded (fp32).  Built from an `XiangqiNet` (ours or a
    loaded reference checkpoint); `forward` has the reference's signature.  Not trainable."""

    def __init__(self, net: XiangqiNet):
        super().__init__()
        self.num_channels = net.num_channels
        self.num_res_blocks = net.num_res_blocks
        self.refresh(net)

    @torch.no_grad()
    def refresh(self, net: XiangqiNet):
        """Re-fold after a weight update (the reference's server.update_model, inference_server.py:476-487)."""
        def reg(name, t):
            t = t.clone().contiguous()
            if name in self._buffers:
                self._buffers[name] = t.to(self._buffers[name].device)
            else:
                self.register_buffer(name, t)

        w, b = _fold(net.input_conv[0].weight, net.input_conv[1])
        reg("w_in", w); reg("b_in", b)
        for i, blk in enumerate(net.res_blocks):
            w1, b1 = _fold(blk.conv1.weight, blk.bn1)
            w2, b2 = _fold(blk.conv2.weight, blk.bn2)
            reg(f"w1_{i}", w1); reg(f"b1_{i}", b1); reg(f"w2_{i}", w2); reg(f"b2_{i}", b2)
        wp, bp = _fold(net.policy_head[0].weight, net.policy_head[1])
        wv, bv = _fold(net.value_head[0].weight, net.value_head[1])
        reg("w_p", wp); reg("b_p", bp); reg("w_v", wv); reg("b_v", bv)
        reg("fc_p_w", net.policy_head[4].weight.detach()); reg("fc_p_b", net.policy_head[4].bias.detach())
        reg("fc_v1_w", net.value_head[4].weight.detach()); reg("fc_v1_b", net.value_head[4].bias.detach())
        reg("fc_v2_w", net.value_head[6].weight.detach()); reg("fc_v2_b", net.value_head[6].bias.detach())

    @torch.no_grad()
    def forward(self, x):
        h = F.relu(F.conv2d(x, self.w_in, self.b_in, padding=1))
        for i in range(self.num_res_blocks):
            y = F.relu(F.conv2d(h, getattr(self, f"w1_{i}"), getattr(self, f"b1_{i}"), padding=1))
            y = F.conv2d(y, getattr(self, f"w2_{i}"), getattr(self, f"b2_{i}"), padding=1)
            h = F.relu(y + h)
        p = F.relu(F.conv2d(h, self.w_p, self.b_p)).flatten(1)
        logits = F.linear(p, self.fc_p_w, self.fc_p_b)
        v = F.relu(F.conv2d(h, self.w_v, self.b_v)).flatten(1)
        v = F.relu(F.linear(v, self.fc_v1_w, self.fc_v1_b))
        value = torch.tanh(F.linear(v, self.fc_v2_w, self.fc_v2_b))
        return logits, value

    def predict(self, state: np.ndarray, device: str = "cpu"):
        with torch.no_grad():
            x = torch.as_tensor(np.asarray(state), dtype=torch.float32).unsqueeze(0).to(self.w_in.device)
            logits, value = self(x)
            probs = F.softmax(logits, dim=1).squeeze(0).cpu().numpy()
        return probs, value.item()


def load_reference_checkpoint(path: str, map_location="cpu") -> XiangqiNet:
    """Read `best_model.pt` / `checkpoint_iter*.pt` as written by the reference (train.py:537-567)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    cfg = ckpt.get("config", {})
    net = XiangqiNet(cfg.get("num_channels", 128), cfg.get("num_res_blocks", 6))
    sd = ckpt.get("best_model_state_dict", ckpt.get("model_state_dict"))
    net.load_state_dict(sd)
    return net
