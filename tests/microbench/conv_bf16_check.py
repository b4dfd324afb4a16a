"""xq_wino_conv3x3_bf16 against the fp32 kernel and a float64 convolution, and its launch time (not a test):
    python tests/microbench/conv_bf16_check.py [B] [C]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.relu(torch.randn(B, 90, C, generator=g)).cuda()
w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
bias = (torch.randn(C, generator=g) * 0.1).cuda()
res = torch.randn(B, 90, C, generator=g).cuda()
u32 = hip.wino_transform_weights(w, 128)
u16 = hip.wino_transform_weights_bf16(w).cuda()
for r in (None, res):
    y32 = torch.full_like(x, float("nan")); y16 = torch.full_like(x, float("nan"))
    hip.wino_conv3x3(x, u32, bias, y32, r, True)
    hip.wino_conv3x3_bf16(x, u16, bias, y16, r, True)
    torch.cuda.synchronize()
    d = (y16 - y32).abs()
    print("residual" if r is not None else "no residual", "nan:", int(torch.isnan(y16).sum()), " max |bf16 - fp32| %.4g  mean %.4g  (output std %.3g)"
          % (d.max().item(), d.mean().item(), y32.std().item()))
    # emulate: same transforms in fp32 but operands rounded to bf16 is hard to restate here; a loose bound: bf16 has 8 bits -> ~1e-2 relative
for name, fn, u in (("fp32", hip.wino_conv3x3, u32), ("bf16", hip.wino_conv3x3_bf16, u16)):
    y = torch.empty_like(x)
    for _ in range(3):
        fn(x, u, bias, y, res, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(x, u, bias, y, res, True)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.4f ms per launch at B=%d C=%d" % (name, e0.elapsed_time(e1) / 20, B, C))
