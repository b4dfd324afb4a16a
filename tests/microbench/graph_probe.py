"""Probe (not a test): per-step latency of small engines with and without a HIP graph around select -> network -> expand."""
import sys, time
import torch
sys.path.insert(0, ".")
from xiangqi_alphazero_amd import engine, evaluator, model, weights

for G, ch, nb in ((1, 128, 6), (16, 128, 6), (256, 128, 6), (1, 256, 10)):
    net = model.XiangqiNet(ch, nb); net.load_state_dict(weights.make_state_dict(ch, nb))
    ev, _ = evaluator.make_evaluator(net, "cuda", "hip")
    eng = engine.SelfPlayEngine(engine.make_config(G, 100, seed=3), evaluator=ev)
    for _ in range(20): eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): eng.step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eng.step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            eng.step()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 200
        st = eng.stats()
        print("G=%d %dx%d: eager %.1f us/step, graph %.1f us/step, overflow %d sims %d" % (G, ch, nb, eager * 1e6, graph * 1e6, st["overflow"], st["sims"]), flush=True)
    except Exception as e:
        print("G=%d %dx%d: eager %.1f us/step, graph capture failed: %r" % (G, ch, nb, eager * 1e6, e), flush=True)
