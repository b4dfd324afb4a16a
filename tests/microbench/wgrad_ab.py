"""A/B of xq_wino_wgrad builds: python tests/microbench/wgrad_ab.py name=path.so ...  (B=256, C=256; time per launch, result check)"""
import ctypes as C, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
B, Cn = 256, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 90, Cn, generator=g).relu().cuda(); dy = (torch.randn(B, 90, Cn, generator=g) * 1e-3).cuda()
x64 = x.double().view(B, 10, 9, Cn).permute(0, 3, 1, 2); d64 = dy.double().view(B, 10, 9, Cn).permute(0, 3, 1, 2)
w = torch.zeros(Cn, Cn, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
F.conv2d(x64, w, None, padding=1).backward(d64)
for spec in sys.argv[1:]:
    name, path = spec.split("=")
    L = C.CDLL(os.path.abspath(path)); vp, i32 = C.c_void_p, C.c_int
    L.xq_wino_wgrad_scratch_bytes.argtypes = [i32, i32]; L.xq_wino_wgrad_scratch_bytes.restype = C.c_size_t
    L.xq_wino_wgrad.argtypes = [vp, vp, vp, vp, i32, i32, vp]; L.xq_wino_wgrad.restype = i32
    scratch = torch.empty(L.xq_wino_wgrad_scratch_bytes(B, Cn) // 4, device="cuda"); dw = torch.empty(Cn, Cn, 3, 3, device="cuda")
    run = lambda: L.xq_wino_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), B, Cn, hip.stream_ptr(x.device))
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    err = (dw.double() - w.grad).abs().max().item() / w.grad.abs().max().item()
    print("%-8s %.1f us per launch (wgrad + reduce)   err/max %.2e" % (name, 1e3 * e0.elapsed_time(e1) / 50, err))
