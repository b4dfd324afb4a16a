"""A/B of xq_stem_conv builds at B = 8192, C = 256 on engine-like planes: python tests/microbench/stem_ab.py name=lib.so ..."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
B, Cn = 8192, 256
g = torch.Generator().manual_seed(1)
x = torch.zeros(B, 15, 90)
idx = torch.randint(0, 14, (B, 90), generator=g); occ = torch.rand(B, 90, generator=g) < 0.35
x.scatter_(1, idx.unsqueeze(1), occ.float().unsqueeze(1)); x[:, 14] = (torch.arange(B) % 2).float().unsqueeze(1)
x = x.view(B, 15, 10, 9).cuda().contiguous()
wt = (torch.randn(135, Cn, generator=g) * 0.1).cuda(); bias = (torch.randn(Cn, generator=g) * 0.1).cuda()
outs = {}
for spec in sys.argv[1:]:
    name, path = spec.split("=")
    L = C.CDLL(os.path.abspath(path)); vp, i32 = C.c_void_p, C.c_int
    L.xq_stem_conv.argtypes = [vp, vp, vp, vp, i32, i32, vp]; L.xq_stem_conv.restype = i32
    y = torch.empty(B, 90, Cn, device="cuda")
    run = lambda: L.xq_stem_conv(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Cn, hip.stream_ptr(x.device))
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    outs[name] = y.clone()
    print("%-6s %.4f ms per launch  (%.2f TB/s of output)" % (name, e0.elapsed_time(e1) / 30, B * 90 * Cn * 4 / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e12))
names = list(outs)
for n in names[1:]:
    print("bitwise %s == %s:" % (names[0], n), torch.equal(outs[names[0]], outs[n]))
