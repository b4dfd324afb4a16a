"""Compare xq_wino_conv3x3 outputs of builds under tests/microbench/lab against the first one (not a test):
    python tests/microbench/cmp_builds.py [--nores] ref_name other_name ...     (names: libxq_<name>.so)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
args = [a for a in sys.argv[1:] if not a.startswith("--")]
nores = "--nores" in sys.argv
B, Cn = 64, 256
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.relu(torch.randn(B, 90, Cn, generator=g)).cuda()
w = (torch.randn(Cn, Cn, 3, 3, generator=g) * (2.0 / (9 * Cn)) ** 0.5).cuda()
u = hip.wino_transform_weights(w, 128); bias = (torch.randn(Cn, generator=g) * 0.1).cuda(); res = torch.randn(B, 90, Cn, generator=g).cuda()
outs = {}
for name in args:
    L = C.CDLL(os.path.abspath('tests/microbench/lab/libxq_%s.so' % name)); vp, i32 = C.c_void_p, C.c_int
    L.xq_wino_conv3x3.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]; L.xq_wino_conv3x3.restype = i32
    y = torch.full_like(x, float('nan'))
    rc = L.xq_wino_conv3x3(x.data_ptr(), u.data_ptr(), bias.data_ptr(), None if nores else res.data_ptr(), y.data_ptr(), B, Cn, 5, hip.stream_ptr(x.device))
    torch.cuda.synchronize(); outs[name] = y
ref = outs[args[0]]
for name, y in outs.items():
    d = (y - ref).abs(); bad = (d > 1e-3)
    idx = bad.nonzero()
    info = ""
    if len(idx):
        pos = idx[:, 1]; ch = idx[:, 2]; row = pos // 9; col = pos % 9
        info = " rows(mod2) %s cols(mod3) %s chquad %s ntile %s boards %d" % (
            sorted(set((row % 2).tolist())), sorted(set((col % 3).tolist())), sorted(set(((ch % 32) // 4).tolist())),
            sorted(set((ch // 32).tolist())), len(set(idx[:, 0].tolist())))
        # is the wrong value the right value without / with another residual?
        gt = idx[:, 0] * 15 + (row // 2) * 3 + col // 3
        info += " tile_in_group %s" % sorted(set((gt % 32).tolist()))
        i0 = idx[0]; info += " first bad %s got %.4f want %.4f res %.4f" % (i0.tolist(), y[tuple(i0)].item(), ref[tuple(i0)].item(), res[tuple(i0)].item())
    print(name, 'max', d.max().item(), 'nan', torch.isnan(y).sum().item(), 'bad', bad.sum().item(), info)
