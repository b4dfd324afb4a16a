import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
B, Cn = 8192, 256
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.relu(torch.randn(B, 90, Cn, generator=g)).cuda(); w = (torch.randn(Cn, Cn, 3, 3, generator=g) * (2.0 / (9 * Cn)) ** 0.5).cuda()
bias = (torch.randn(Cn, generator=g) * 0.1).cuda(); res = torch.randn(B, 90, Cn, generator=g).cuda(); u16 = hip.wino_transform_weights_bf16(w).cuda(); y = torch.empty_like(x)
for name in sys.argv[1:]:
    L = C.CDLL(os.path.abspath("tests/microbench/lab/libxq_%s.so" % name)); vp, i32 = C.c_void_p, C.c_int
    L.xq_wino_conv3x3_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]; L.xq_wino_conv3x3_bf16.restype = i32
    run = lambda: L.xq_wino_conv3x3_bf16(x.data_ptr(), u16.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, Cn, 1, hip.stream_ptr(x.device))
    for _ in range(3): run()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize(); print("%-8s %.4f ms" % (name, e0.elapsed_time(e1) / 20))
