"""Where does the fp32 Winograd tower leave the 1e-5 contract?  (GPU; writes one JSON object to stdout.)

    python tools/measure_winograd_margin.py > profiles/r03_winograd_margin.json

The 256x20 peaked generator network is re-parameterised without changing its function (weights.rescale_channels): activation
scales spread log-uniformly over 10^-D .. 10^+D per channel, optionally a few residual-stream channels at magnitude `big`,
and the hand-written evaluator is compared with the float64 evaluation of the same folded weights (the helper of
tests/test_nn_fullsize.py).  The bound the tests assert (D <= 2, four channels at 1e3) is the documented domain of validity."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import test_nn_fullsize as T  # noqa: E402

rows = []
for ch, nb, gain in ((256, 20, 1.0), (256, 20, 8.0), (256, 10, 1.0)):
    for decades, nbig, big in ((0, 0, 1e3), (1, 0, 1e3), (2, 0, 1e3), (3, 0, 1e3), (4, 0, 1e3), (6, 0, 1e3),
                               (2, 4, 1e3), (2, 4, 1e4), (2, 4, 1e5), (2, 4, 1e6), (0, 16, 1e4)):
        r = T.winograd_margin_case(ch, nb, gain, float(decades), nbig, big)
        rows.append({"net": "%dx%d" % (ch, nb), "policy_gain": gain, "scale_spread_decades": decades, "big_channels": nbig,
                     "big_magnitude": big if nbig else None, "hip_winograd": r["hip"], "torch_fp32_direct": r["torch"],
                     "largest_activation_seen": r["largest_activation"], "max_abs_logit": r["max_abs_logit"],
                     "hip_inside_1e-5": bool(r["hip"]["prob"] < 1e-5 and r["hip"]["value"] < 1e-5)})
        print(rows[-1], file=sys.stderr, flush=True)
json.dump({"what": "hand-written fp32 evaluator (Winograd F(2,3)xF(3,3) tower) and the plain PyTorch fp32 network vs float64 of the same "
                   "folded weights, 48 corpus boards; prob / value = max abs error, logit_rel = max abs logit error / max |logit|",
           "rows": rows}, sys.stdout, indent=1)
