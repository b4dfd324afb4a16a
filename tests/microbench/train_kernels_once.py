"""The train-step kernels at the train step's own size (batch 256, C = 256), a few launches each, for profiler passes:
    rocprofv3 --pmc <counters> --kernel-trace -f csv -d out -- python3 tests/microbench/train_kernels_once.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip, native_conv
B, C = 256, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, 10, 9, generator=g).relu().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda().requires_grad_(True)
bn = torch.nn.BatchNorm2d(C).cuda().train()
gy = (torch.randn(B, C, 10, 9, generator=g) * 1e-3).cuda().contiguous(memory_format=torch.channels_last)
for _ in range(6):
    y = native_conv.bn_act(native_conv.conv3x3(x, w), bn, x, True)
    y.backward(gy)
    x.grad = None; w.grad = None
torch.cuda.synchronize()
print("done")
