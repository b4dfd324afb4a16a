#!/usr/bin/env python3
"""Build-container only (needs /root/reference): times the reference's own MCTS.search (its Python tree + Cython engine +
XiangqiNet.predict, one thread) beside the CPU port that bench.py uses as cpu_baseline (C search of oracle/ + the same
batch-1 predict), BASELINE configs[0]: 1 game, 100 sims/move, 128ch x 6blk.  Shows the port is not slower than the
original (calibration)."""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = ["/root/reference/training", os.path.join(ROOT, "oracle", "_ref"), ROOT]

import torch  # noqa: E402

torch.set_num_threads(1)
import game as ref_game  # noqa: E402
import mcts as ref_mcts  # noqa: E402
import model as ref_model  # noqa: E402

from oracle import xq_oracle as O  # noqa: E402
from xiangqi_alphazero_amd import weights  # noqa: E402

assert ref_game._USE_CYTHON
net = ref_model.XiangqiNet(num_channels=128, num_res_blocks=6)
net.load_state_dict(weights.make_state_dict(128, 6))
net.eval()
sims = 100
g = ref_game.XiangqiGame()
m = ref_mcts.MCTS(net, num_simulations=sims, c_puct=1.5, device="cpu")
m.search(g, temperature=1.0, add_noise=False)            # warm-up
t0 = time.perf_counter()
pi_ref = m.search(g, temperature=1.0, add_noise=False)
t_ref = time.perf_counter() - t0
og = O.Game()
O.mcts_search(og, 8, lambda s: net.predict(s, "cpu"))
t0 = time.perf_counter()
r = O.mcts_search(og, sims, lambda s: net.predict(s, "cpu"))
t_port = time.perf_counter() - t0
same = list(r.visits[:r.n_children]) == [int(round(pi_ref[a] * sims)) for a in r.actions[:r.n_children]]
print(json.dumps({"config": "1 game, 100 sims/move, 128x6, 1 thread", "reference_sims_per_s": round(sims / t_ref, 1),
                  "port_sims_per_s": round(sims / t_port, 1), "same_visit_counts": bool(same)}))
