import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
C, B = int(sys.argv[1]), int(sys.argv[2])
gen = torch.Generator().manual_seed(1)
x = torch.randn(B, 90, C, generator=gen).cuda(); dy = torch.randn(B, 90, C, generator=gen).cuda()
dw = hip.wino_wgrad(x, dy).double()
x64 = x.double().view(B, 10, 9, C).permute(0, 3, 1, 2); dy64 = dy.double().view(B, 10, 9, C).permute(0, 3, 1, 2)
w = torch.zeros(C, C, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
F.conv2d(x64, w, None, padding=1).backward(dy64)
err = (dw - w.grad).abs()
print("max err", err.max().item(), "max ref", w.grad.abs().max().item())
print("err by (r,s):\n", err.amax(dim=(0, 1)))
e2 = err.amax(dim=(2, 3))
print("err by co block of 32 x ci block of 32:\n", e2.view(C // 32, 32, C // 32, 32).amax(dim=(1, 3)))
print("err by co%32 (first block):", e2[:32, :32].amax(dim=1))
print("err by ci%32 (first block):", e2[:32, :32].amax(dim=0))
