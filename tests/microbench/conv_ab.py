"""Interleaved A/B timing of xq_wino_conv3x3 builds in ONE process (not a test):
    python tests/microbench/conv_ab.py [--b 8192] [--c 256] [--wide 1] [--rounds 7] [--iters 20] name=path.so [name=path.so ...]
Every library is loaded side by side with ctypes; each round times `iters` back-to-back launches of every build in turn
(HIP events on torch's current stream), so box-to-box and clock drift hit all builds alike.  Prints median / min per build
and checks that the builds agree bit for bit on the first output (`--check 0` for ablation builds, whose results are wrong)."""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip  # noqa: E402  (weight transform + stream pointer only)

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=8192)
ap.add_argument("--c", type=int, default=256)
ap.add_argument("--wide", type=int, default=1)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--residual", type=int, default=1)
ap.add_argument("--relu_input", type=int, default=1, help="post-ReLU activations like the tower's (1) or dense randn (0)")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()

B, Cn = a.b, a.c
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(B, 90, Cn, generator=g).cuda()
if a.relu_input:
    x = torch.relu(x)
w = (torch.randn(Cn, Cn, 3, 3, generator=g) * (2.0 / (9 * Cn)) ** 0.5).cuda()
u = hip.wino_transform_weights(w, 128 if a.wide else 64)
bias = (torch.randn(Cn, generator=g) * 0.1).cuda()
res = torch.randn(B, 90, Cn, generator=g).cuda()
flags = 1 | (4 if a.wide else 0)
stream = hip.stream_ptr(x.device)

libs = []
for spec in a.libs:
    name, path = spec.split("=", 1)
    extra = 0
    if "+" in path:                                   # name=path.so+EXTRAFLAGS (e.g. +2056 = XQ_CONV_STAGGER | 8 << 8)
        path, ex = path.rsplit("+", 1)
        extra = int(ex)
    L = C.CDLL(os.path.abspath(path))
    vp, i32 = C.c_void_p, C.c_int
    L.xq_wino_conv3x3.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.xq_wino_conv3x3.restype = i32
    libs.append((name, L, torch.full_like(x, float("nan")), extra))


def run(L, y, extra=0):
    rc = L.xq_wino_conv3x3(x.data_ptr(), u.data_ptr(), bias.data_ptr(), res.data_ptr() if a.residual else None, y.data_ptr(), B, Cn,
                           flags | extra, stream)
    if rc != 0:
        raise RuntimeError("xq_wino_conv3x3 -> %d" % rc)


for name, L, y, ex in libs:
    for _ in range(3):
        run(L, y, ex)
torch.cuda.synchronize()
if a.check:
    for name, L, y, ex in libs[1:]:
        same = torch.equal(libs[0][2], y)
        print("bitwise %s == %s: %s   max|d| %.3g" % (libs[0][0], name, same, (libs[0][2] - y).abs().max().item()))
times = {name: [] for name, _, _, _ in libs}
for r in range(a.rounds):
    for name, L, y, ex in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run(L, y, ex)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / a.iters)
wino = 20 * 2.0 * ((B * 15 + 31) // 32 * 32) * Cn * Cn
for name, _, _, _ in libs:
    t = times[name]
    med, mn = statistics.median(t), min(t)
    print("%-12s B=%d C=%d wide=%d  median %.4f ms  min %.4f ms  (%.1f%% of 157.3 TF at the median)  all: %s"
          % (name, B, Cn, a.wide, med, mn, wino / med / 1e9 / 157.3 * 100, " ".join("%.3f" % v for v in t)))
