"""Counter-based deterministic weights for the policy/value ResNet.

There are no shipped checkpoints in the reference (models/*.pt is git-ignored) and no network here,
so benchmarks and NN parity fixtures use weights produced by this generator: a pure function of
(tensor name, element index, seed) built from 64-bit integer mixing, identical on every host.
Keys and shapes are those of the reference `XiangqiNet.state_dict()` (training/model.py:48-85), so the
result loads into the reference model with `load_state_dict` as well as into ours.
"""
from __future__ import annotations

import re
import zlib
from collections import OrderedDict

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _uniform(name: str, n: int, seed: int) -> np.ndarray:
    """n float64 values in [-0.5, 0.5), exact multiples of 2^-24."""
    key = np.uint64((zlib.crc32(name.encode()) << 32) ^ (seed & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _M1 + key
        x = (x ^ (x >> np.uint64(30))) * _M2
        x = (x ^ (x >> np.uint64(27))) * _M3
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5


def state_dict_shapes(num_channels: int, num_res_blocks: int) -> "OrderedDict[str, tuple]":
    """Key -> shape, in the order torch enumerates the reference module."""
    c = num_channels
    d: "OrderedDict[str, tuple]" = OrderedDict()

    def bn(prefix, ch):
        d[prefix + ".weight"] = (ch,)
        d[prefix + ".bias"] = (ch,)
        d[prefix + ".running_mean"] = (ch,)
        d[prefix + ".running_var"] = (ch,)
        d[prefix + ".num_batches_tracked"] = ()

    d["input_conv.0.weight"] = (c, 15, 3, 3)
    bn("input_conv.1", c)
    for i in range(num_res_blocks):
        d[f"res_blocks.{i}.conv1.weight"] = (c, c, 3, 3)
        bn(f"res_blocks.{i}.bn1", c)
        d[f"res_blocks.{i}.conv2.weight"] = (c, c, 3, 3)
        bn(f"res_blocks.{i}.bn2", c)
    d["policy_head.0.weight"] = (32, c, 1, 1)
    bn("policy_head.1", 32)
    d["policy_head.4.weight"] = (8100, 2880)
    d["policy_head.4.bias"] = (8100,)
    d["value_head.0.weight"] = (4, c, 1, 1)
    bn("value_head.1", 4)
    d["value_head.4.weight"] = (128, 360)
    d["value_head.4.bias"] = (128,)
    d["value_head.6.weight"] = (1, 128)
    d["value_head.6.bias"] = (1,)
    return d


def make_state_dict_numpy(num_channels: int, num_res_blocks: int, seed: int = 0,
                          policy_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """He-style scaled uniform weights; BN statistics close to identity but not trivial."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in state_dict_shapes(num_channels, num_res_blocks).items():
        n = int(np.prod(shape)) if shape else 1
        if name.endswith("num_batches_tracked"):
            out[name] = np.array(1, dtype=np.int64)
            continue
        u = _uniform(name, n, seed)
        is_bn = bool(re.search(r"(\.bn[12]|input_conv\.1|_head\.1)\.", name))
        if name.endswith("running_var"):
            v = 1.0 + 0.4 * u                       # 0.8 .. 1.2
        elif name.endswith("running_mean"):
            v = 0.2 * u
        elif is_bn:
            v = (1.0 + 0.2 * u) if name.endswith("weight") else 0.1 * u
        elif name.endswith(".bias"):
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0
            if name == "policy_head.4.weight":
                gain = policy_gain
            elif name.endswith("conv2.weight"):
                gain = 0.3        # keeps the residual tower's activations O(1) at any depth
            elif name == "value_head.6.weight":
                gain = 0.5
            v = u * 2.0 * np.sqrt(3.0 / fan_in) * np.sqrt(2.0) * gain
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def make_state_dict(num_channels: int, num_res_blocks: int, seed: int = 0, policy_gain: float = 1.0):
    """Same as make_state_dict_numpy but as torch tensors (torch imported lazily)."""
    import torch
    sd = OrderedDict()
    for k, v in make_state_dict_numpy(num_channels, num_res_blocks, seed, policy_gain).items():
        sd[k] = torch.from_numpy(np.ascontiguousarray(v)) if v.shape else torch.tensor(int(v), dtype=torch.long)
    return sd


# crc32 over make_state_dict_numpy(64, 3) in key order; pins the generator (fixtures depend on it)
REFERENCE_CRC_64x3 = 0xE8FEBBD9


def rescale_channels(sd, num_res_blocks: int, decades: float, seed: int = 0, big_channels: int = 0, big: float = 1.0e3):
    """A function-preserving re-parameterisation of a `make_state_dict` result that spreads the activation scales, for the
    numerical-margin test of the Winograd tower (tests/test_nn_fullsize.py): residual-stream channel c is multiplied by
    S_c and every block's inner channel c by T_c^(i), S and T log-uniform over [10^-decades, 10^+decades] (`big_channels`
    stream channels get S_c = `big` instead), by scaling the producing BatchNorm's weight and bias and dividing the
    consuming convolution's input-channel weights.  ReLU is positively homogeneous and the skip connection adds equally
    scaled channels, so in exact arithmetic the network's outputs do not change; in float32 the operand dynamic range of
    every 3x3 convolution does.  Returns a new OrderedDict of torch tensors (float64 arithmetic, stored float32)."""
    import torch
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    c = out["input_conv.0.weight"].shape[0]

    def scales(tag):
        u = _uniform("rescale." + tag, c, seed) * 2.0                     # [-1, 1)
        return torch.from_numpy(10.0 ** (decades * u))

    def scale_bn(prefix, f):
        for k in (".weight", ".bias"):
            out[prefix + k] = (out[prefix + k].double() * f).float()

    def scale_in(key, f):                                                # divide the input-channel axis
        out[key] = (out[key].double() / f.view(1, -1, 1, 1)).float()

    s = scales("stream")
    if big_channels > 0:
        s[:big_channels] = big
    scale_bn("input_conv.1", s)
    for i in range(num_res_blocks):
        t = scales("inner%d" % i)
        scale_in(f"res_blocks.{i}.conv1.weight", s)
        scale_bn(f"res_blocks.{i}.bn1", t)
        scale_in(f"res_blocks.{i}.conv2.weight", t)
        scale_bn(f"res_blocks.{i}.bn2", s)
    scale_in("policy_head.0.weight", s)
    scale_in("value_head.0.weight", s)
    return out
