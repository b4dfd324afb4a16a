"""Float32 emulation of the residual tower under different Winograd tilings, against the float64 tower (CPU, no GPU needed):

    python tools/winograd_margin.py [--channels 256] [--blocks 20] [--gain 8] [--boards 8]

Answers two questions with numbers:
  * the row tiling F(5,3) (2 tiles x 7 frequencies: 210 instead of 300 multiplies per board and channel pair) -- is it inside
    the 1e-5 contract with margin?  (It is what a further cut of the MFMA work would need.)
  * the tiling the kernel uses, F(2,3) rows x F(3,3) columns -- how far from the contract is it?
Every product is float32 with float32 accumulation over the input channels (numpy matmul), the transforms are float32 with
the kernel's scaling (column B^T rows scaled to small integers, inverse scales folded into the float64-computed weights).
Interpolation points: F(2,3): 0, +-1, inf; F(3,3): 0, +-1, 2, inf; F(5,3): 0, +-1, +-2, 1/2, inf."""
import argparse
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cook_toom(m, r, pts):
    """Exact (Fraction) A^T [m x n], G [n x r], B^T [n x n] of F(m, r) for finite points `pts` (n - 1 of them) + infinity."""
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fraction(p) for p in pts]
    AT = [[(pts[j] ** i if j < n - 1 else Fraction(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        nj = Fraction(1)
        for l in range(n - 1):
            if l != j:
                nj *= pts[j] - pts[l]
        G.append([pts[j] ** k / nj for k in range(r)])
    G.append([Fraction(int(k == r - 1)) for k in range(r)])
    # B^T: row j < n-1 = coefficients of prod_{l != j} (x - a_l); last row = coefficients of prod_l (x - a_l)
    def poly_mul(p, q):
        out = [Fraction(0)] * (len(p) + len(q) - 1)
        for i, a in enumerate(p):
            for k, b in enumerate(q):
                out[i + k] += a * b
        return out
    BT = []
    for j in range(n):
        p = [Fraction(1)]
        for l in range(n - 1):
            if l != j:
                p = poly_mul(p, [-pts[l], Fraction(1)])
        BT.append(p + [Fraction(0)] * (n - len(p)))
    f = lambda M: np.array([[float(x) for x in row] for row in M], dtype=np.float64)
    AT, G, BT = f(AT), f(G), f(BT)
    # self-check on random data (float64)
    rs = np.random.RandomState(0)
    d, g = rs.randn(n), rs.randn(r)
    want = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    got = AT @ ((G @ g) * (BT @ d))
    assert np.allclose(got, want, atol=1e-9), (got, want)
    return AT, G, BT


def make_tiling(rows_m, rows_pts, cols_m, cols_pts):
    return cook_toom(rows_m, 3, rows_pts), cook_toom(cols_m, 3, cols_pts)


def wino_conv_f32(x, w, bias, tiling, H=10, W=9):
    """x float32 [B, C, H, W] (zero padding 1), w float64 [Co, Ci, 3, 3]; every transform and product in float32."""
    (ATr, Gr, BTr), (ATc, Gc, BTc) = tiling
    mr, nr = ATr.shape
    mc, nc = ATc.shape
    B, C = x.shape[:2]
    # scale B^T rows to small integers where possible and fold the inverse scale into G (computed in float64, stored float32)
    def integerise(BT, G):
        BT, G = BT.copy(), G.copy()
        for j in range(BT.shape[0]):
            row = BT[j]
            for s in (1, 2, 4, 6, 3, 8, 12):
                if np.allclose(row * s, np.round(row * s), atol=1e-12):
                    BT[j] = row * s
                    G[j] = G[j] / s
                    break
        return BT, G
    BTr_i, Gr_i = integerise(BTr, Gr)
    BTc_i, Gc_i = integerise(BTc, Gc)
    U = np.einsum("pa,oiab,qb->pqoi", Gr_i, w.astype(np.float64), Gc_i).astype(np.float32)      # [nr, nc, Co, Ci]
    tr, tc = -(-H // mr), -(-W // mc)
    xp = np.zeros((B, C, tr * mr + 2, tc * mc + 2), dtype=np.float32)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    y = np.zeros((B, w.shape[0], tr * mr, tc * mc), dtype=np.float32)
    BTr32, BTc32, ATr32, ATc32 = (m.astype(np.float32) for m in (BTr_i, BTc_i, ATr, ATc))
    for ty in range(tr):
        for tx in range(tc):
            d = xp[:, :, ty * mr:ty * mr + nr, tx * mc:tx * mc + nc]                       # [B, C, nr, nc]
            v = np.einsum("pa,bcaq->bcpq", BTr32, d).astype(np.float32)
            v = np.einsum("qe,bcpe->bcpq", BTc32, v).astype(np.float32)                    # [B, C, nr, nc]
            m_ = np.empty((B, w.shape[0], nr, nc), dtype=np.float32)
            for p in range(nr):
                for q in range(nc):
                    m_[:, :, p, q] = v[:, :, p, q] @ U[p, q].T                              # float32 matmul over Ci
            o = np.einsum("ap,bopq->boaq", ATr32, m_).astype(np.float32)
            o = np.einsum("eq,boaq->boae", ATc32, o).astype(np.float32)
            y[:, :, ty * mr:(ty + 1) * mr, tx * mc:(tx + 1) * mc] = o
    return y[:, :, :H, :W] + bias.astype(np.float32)[None, :, None, None]


def direct_conv(x, w, bias, dtype):
    import torch
    import torch.nn.functional as F
    t = torch.float64 if dtype == np.float64 else torch.float32
    return F.conv2d(torch.from_numpy(x).to(t), torch.from_numpy(w).to(t), torch.from_numpy(bias).to(t), padding=1).numpy()


def tower(x, inf, conv, blocks):
    h = np.maximum(direct_conv(x, inf["w_in"], inf["b_in"], x.dtype.type), 0)
    for i in range(blocks):
        y = np.maximum(conv(h, inf[f"w1_{i}"], inf[f"b1_{i}"]), 0)
        y = conv(y, inf[f"w2_{i}"], inf[f"b2_{i}"])
        h = np.maximum(y + h, 0)
    return h


def heads(h, inf):
    import torch
    import torch.nn.functional as F
    t = torch.from_numpy(h).double()
    dd = lambda a: torch.from_numpy(np.asarray(a)).double()
    p = F.relu(F.conv2d(t, dd(inf["w_p"]), dd(inf["b_p"]))).flatten(1)
    logits = p @ dd(inf["fc_p_w"]).t() + dd(inf["fc_p_b"])
    v = F.relu(F.conv2d(t, dd(inf["w_v"]), dd(inf["b_v"]))).flatten(1)
    v = F.relu(v @ dd(inf["fc_v1_w"]).t() + dd(inf["fc_v1_b"]))
    v = torch.tanh(v @ dd(inf["fc_v2_w"]).t() + dd(inf["fc_v2_b"]))
    return torch.softmax(logits, 1).numpy(), v.numpy().reshape(-1), logits.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--gain", type=float, default=8.0)
    ap.add_argument("--boards", type=int, default=8)
    a = ap.parse_args()
    import torch
    import golden_io as G
    from oracle import xq_oracle as O
    from xiangqi_alphazero_amd import model, weights
    torch.set_num_threads(8)
    net = model.XiangqiNet(a.channels, a.blocks)
    net.load_state_dict(weights.make_state_dict(a.channels, a.blocks, policy_gain=a.gain))
    ref = model.InferenceNet(net)
    names = ["w_in", "b_in", "w_p", "b_p", "w_v", "b_v", "fc_p_w", "fc_p_b", "fc_v1_w", "fc_v1_b", "fc_v2_w", "fc_v2_b"]
    names += [f"{k}{j}_{i}" for i in range(a.blocks) for j in (1, 2) for k in ("w", "b")]
    inf = {n: getattr(ref, n).detach().double().numpy() for n in names}
    d = G.corpus()
    idx = np.linspace(0, len(d["board"]) - 1, a.boards).astype(int)
    x = np.stack([O.encode_state(d["board"][i], int(d["side"][i])) for i in idx]).astype(np.float32)

    p64, v64, l64 = heads(tower(x.astype(np.float64), inf, lambda h, w, b: direct_conv(h, w, b, np.float64), a.blocks), inf)
    tilings = {
        "direct float32": None,
        "F(2,3) rows x F(3,3) cols (the kernel)": make_tiling(2, [0, 1, -1], 3, [0, 1, -1, 2]),
        "F(5,3) rows x F(3,3) cols": make_tiling(5, [0, 1, -1, 2, -2, Fraction(1, 2)], 3, [0, 1, -1, 2]),
    }
    print("%dx%d, policy_gain %g, %d boards; max |p - p64|, max |v - v64|, max |logit - logit64| / max|logit64|; contract 1e-5"
          % (a.channels, a.blocks, a.gain, a.boards))
    for name, tl in tilings.items():
        if tl is None:
            conv = lambda h, w, b: direct_conv(h.astype(np.float32), w, b, np.float32)
        else:
            conv = lambda h, w, b, tl=tl: wino_conv_f32(h.astype(np.float32), w, b, tl)
        h = tower(x, inf, conv, a.blocks)
        p, v, l = heads(h, inf)
        print("  %-42s  probs %.3g   value %.3g   logits %.3g (top prob %.3f)" % (
            name, np.abs(p - p64).max(), np.abs(v - v64).max(), np.abs(l - l64).max() / np.abs(l64).max(), p64.max()), flush=True)


if __name__ == "__main__":
    main()
