"""Diagnostic: per-block phase timestamps of k_wino_conv (not a test)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from xiangqi_alphazero_amd import hip

B, Cc = 8192, 256
x = torch.randn(B, 90, Cc, device="cuda"); w = torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.02
u = hip.wino_transform_weights(w); bias = torch.randn(Cc, device="cuda"); res = torch.randn(B, 90, Cc, device="cuda")
y = torch.empty_like(x)
grid = 12800
st = torch.zeros(grid * 16, dtype=torch.int64, device="cuda")
L = hip.lib()
for _ in range(3):
    rc = L.xq_wino_conv3x3_dbg(x.data_ptr(), u.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, Cc, 1, st.data_ptr(), hip.stream_ptr(x.device))
    assert rc == 0
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(grid, 16).astype(np.int64)
t0 = s[:, 0].min()
pro, main, epi = (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0, (s[:, 3] - s[:, 2]) / 100.0     # us
print("blocks %d  span %.1f us" % (grid, (s[:, 3].max() - t0) / 100.0))
print("prologue us  mean %.2f p50 %.2f p95 %.2f" % (pro.mean(), np.median(pro), np.percentile(pro, 95)))
print("main     us  mean %.2f p50 %.2f p95 %.2f  (per chunk %.3f)" % (main.mean(), np.median(main), np.percentile(main, 95), main.mean() / 32))
print("epilogue us  mean %.2f p50 %.2f p95 %.2f" % (epi.mean(), np.median(epi), np.percentile(epi, 95)))
hw = s[:, 8:16]
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
print("wave -> SIMD id of the first blocks:", simd[:6].tolist())
print("all waves of a block on one CU:", bool((cu == cu[:, :1]).all()), " partners (w, w+4) share a SIMD in %.1f%% of blocks" % (100.0 * (simd[:, :4] == simd[:, 4:]).all(axis=1).mean()))
print("partners (2k, 2k+1) share a SIMD in %.1f%% of blocks" % (100.0 * (simd[:, 0::2] == simd[:, 1::2]).all(axis=1).mean()))
