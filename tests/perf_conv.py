"""Micro-benchmark of xq_wino_conv3x3 (not a test): python tests/perf_conv.py [B] [C]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xiangqi_alphazero_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = torch.randn(B, 90, C, device="cuda")
w = torch.randn(C, C, 3, 3, device="cuda") * (2.0 / (9 * C)) ** 0.5
CB = int(os.environ.get('XQ_CONV_BLOCK', '64'))
u = hip.wino_transform_weights(w, CB)
bias = torch.randn(C, device="cuda")
res = torch.randn(B, 90, C, device="cuda")
y = torch.empty_like(x)
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 1


def run():
    hip.check(hip.lib().xq_wino_conv3x3(x.data_ptr(), u.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, C,
                                        FLAGS | (4 if CB == 128 else 0), hip.stream_ptr(x.device)), "conv")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
direct = 2.0 * B * 90 * 9 * C * C
wino = 20 * 2.0 * ((B * 15 + 31) // 32 * 32) * C * C
print("flags=%d " % FLAGS + "B=%d C=%d  %.3f ms  direct-equivalent %.1f TFLOP/s  MFMA (winograd flops) %.1f TFLOP/s = %.1f%% of 157.3"
      % (B, C, ms, direct / ms / 1e9, wino / ms / 1e9, wino / ms / 1e9 / 157.3 * 100))
