import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import hip
C, B = 64, 1
for (co0, ci0, pos) in [(0, 0, 40), (1, 0, 40), (0, 1, 40), (5, 9, 40), (33, 2, 13), (2, 35, 77)]:
    x = torch.zeros(B, 90, C).cuda(); dy = torch.zeros(B, 90, C).cuda()
    x[0, pos, ci0] = 1.0; dy[0, pos, co0] = 1.0
    dw = hip.wino_wgrad(x, dy)
    nz = (dw.abs() > 1e-4).nonzero().tolist()
    print((co0, ci0, pos), "->", [(a, b, c, d, round(dw[a, b, c, d].item(), 3)) for a, b, c, d in nz][:8])
