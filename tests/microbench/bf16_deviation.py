"""Measured deviation of the reduced-precision evaluators (kinds 'bf16' = hand-written bf16 convolution, 'bf16-lib' = ROCm library,
all-bf16) from the float32 product evaluator on the same positions: value, logits (relative to the row's largest |logit|),
probabilities (absolute and relative on the top moves).  python tests/microbench/bf16_deviation.py [channels blocks gain]"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import engine, evaluator, model, weights

out = {}
for (c, b, gain) in [(128, 6, 1.0), (256, 10, 1.0), (256, 20, 8.0)]:
    net = model.XiangqiNet(c, b)
    net.load_state_dict(weights.make_state_dict(c, b, policy_gain=gain))
    e32, _ = evaluator.make_evaluator(net, "cuda", "hip")
    eng = engine.SelfPlayEngine(engine.make_config(512, 16, random_opening_moves=12, seed=7), "cuda", evaluator=e32)
    for _ in range(40):
        eng.step()
    x = eng.select().clone()
    l32, v32 = e32(x, full_policy=True)
    l32, v32 = l32.double(), v32.double()
    p32 = torch.softmax(l32, 1)
    row = {}
    for kind in ("bf16", "bf16-lib"):
        ev, name = evaluator.make_evaluator(net, "cuda", kind)
        l, v = ev(x, full_policy=True) if kind == "bf16" else ev(x)
        l, v = l.double(), v.double()
        p = torch.softmax(l, 1)
        top = p32.topk(5, dim=1)
        row[kind] = {"evaluator": name,
                     "value_max_abs": float((v - v32).abs().max()), "value_mean_abs": float((v - v32).abs().mean()),
                     "logit_max_rel_to_row_max": float(((l - l32).abs().max(1).values / l32.abs().max(1).values).max()),
                     "prob_max_abs": float((p - p32).abs().max()),
                     "top5_prob_max_rel": float(((p.gather(1, top.indices) - top.values).abs() / top.values).max()),
                     "top1_move_agrees": float((p.argmax(1) == p32.argmax(1)).double().mean())}
    out["%dx%d policy_gain %g" % (c, b, gain)] = row
print(json.dumps(out, indent=1))
