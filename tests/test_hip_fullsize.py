"""Size-independent properties of the self-play engine at BASELINE.json's full size (configs[2]: 8192 concurrent games,
800 simulations per move), where the CPU oracle cannot follow: the trees and the samples must obey the invariants of
mcts.py / parallel_selfplay.py for every slot.  The evaluator is a device-side pseudo-network (one fp32 GEMM of the
planes with a fixed random matrix), so 800+ steps take seconds."""
import numpy as np
import pytest

import golden_io as G

pytestmark = pytest.mark.gpu

GAMES, SIMS = 8192, 800


class PseudoNet:
    """logits = planes . R (R fixed, seeded), value = tanh of a second projection: position-dependent, cheap."""

    def __init__(self, device):
        import torch
        g = torch.Generator(device="cpu").manual_seed(1234)
        self.r = (torch.randn(1350, 8100, generator=g) * 0.35).to(device)
        self.v = (torch.randn(1350, generator=g) * 0.2).to(device)

    def __call__(self, x):
        import torch
        f = x.reshape(x.shape[0], 1350)
        return f @ self.r, torch.tanh(f @ self.v)


def _children_sum(t, first, cnt, ncols):
    """sum of t over [first, first+cnt) per node, via an exclusive prefix sum along the node axis."""
    import torch
    cs = torch.zeros((t.shape[0], ncols + 1), dtype=torch.float64 if t.dtype.is_floating_point else torch.int64, device=t.device)
    torch.cumsum(t[:, :ncols], dim=1, out=cs[:, 1:])
    # entries beyond a slot's allocation are whatever the (recycled) workspace held: clamp both ends, the callers mask them
    lo = first.clamp(min=0, max=ncols).long()
    hi = (lo + cnt.long()).clamp(max=ncols)
    return torch.gather(cs, 1, hi) - torch.gather(cs, 1, lo)


def _evaluator(kind):
    """'pseudo-dense': the device-side pseudo-network through the dense [G,8100] hand-off (xq_engine_expand);
    'hip-sparse': the hand-written ResNet evaluator (64x3, peaked policy) through the product hand-off -- legal-move logits
    [G,128] from xq_policy_head_legal into xq_engine_expand_legal."""
    if kind == "pseudo-dense":
        return PseudoNet("cuda")
    from xiangqi_alphazero_amd import evaluator, model, weights
    net = model.XiangqiNet(64, 3)
    net.load_state_dict(weights.make_state_dict(64, 3, policy_gain=4.0))
    return evaluator.make_evaluator(net, "cuda", "hip")[0]


@pytest.mark.parametrize("kind", ["pseudo-dense", "hip-sparse"])
def test_search_trees_at_full_size(kind):
    import torch
    from xiangqi_alphazero_amd import engine, hip
    d = G.corpus()
    picks = [i for i in range(len(d["board"])) if not d["done"][i]]
    eng = engine.SelfPlayEngine(engine.make_config(GAMES, SIMS, add_noise=False, manual_moves=True), evaluator=_evaluator(kind))
    boards = np.stack([d["board"][picks[s % len(picks)]] for s in range(GAMES)])
    sides = np.array([d["side"][picks[s % len(picks)]] for s in range(GAMES)], dtype=np.int8)
    for s in range(GAMES):
        i = picks[s % len(picks)]
        eng.set_position(s, boards[s], int(sides[s]), int(d["move_count"][i]), int(d["no_capture"][i]), G.history_tail(d, i))
    for _ in range(SIMS + 1):
        eng.step()
    st = eng.stats()
    assert st["overflow"] == 0
    assert st["sims"] == GAMES * SIMS and st["root_evals"] == GAMES
    assert st["leaf_evals"] + st["terminal_sims"] == st["sims"]
    ints = eng.slot_ints.cpu().numpy()
    assert (ints[:, 4] == SIMS).all()                               # simulations done, every slot
    alloc = torch.from_numpy(ints[:, 7].astype(np.int64)).cuda()     # nodes allocated per slot
    av = eng.arena_views()
    ncols = int(alloc.max().item())
    assert ncols <= av["node_cap"] and int(alloc.sum().item()) - GAMES == st["nodes_created"]
    N, P, first = av["N"][:, :ncols], av["P"][:, :ncols], av["first"][:, :ncols]
    meta = av["meta"][:, :ncols].int() & 0xFFFF
    cnt, kind = meta & 0x3FFF, meta >> 14
    idx = torch.arange(ncols, device="cuda")[None, :]
    live = idx < alloc[:, None]
    expanded = live & (cnt > 0)
    # structure: children of an expanded node are a run inside the slot's allocation, unexpanded nodes have none
    assert bool(((first >= 1) & (first + cnt <= alloc[:, None]))[expanded].all())
    assert bool((first[live & (cnt == 0)] == -1).all())
    assert int(cnt[expanded].sum().item()) == int(alloc.sum().item()) - GAMES      # every non-root node is someone's child
    # visit counts (mcts.py:126-153): a simulation adds one visit to every node of its path, so an expanded node has
    # its own expanding visit plus its children's; the root was expanded by the root evaluation, not by a simulation
    child_n = _children_sum(N, first, cnt, ncols)
    own = (idx > 0).long()
    assert bool((N.long() == child_n + own)[expanded].all())
    assert bool((N[:, 0] == SIMS).all()) and bool((N[live] >= 0).all())
    # |W| <= N: every backed-up value is in [-1, 1]
    assert bool((av["W"][:, :ncols].abs() <= N.double() + 1e-9)[live].all())
    # float32 priors of an expanded (noise-free) node are a distribution over its children
    psum = _children_sum(P, first, cnt, ncols)
    assert bool(((psum - 1.0).abs() < 2e-5)[expanded & (kind == 0)].all())
    assert bool((P[live] >= 0).all())
    # root children are the ordered legal moves of the root position
    mv, cn, _, _ = hip.movegen(torch.from_numpy(boards).cuda(), torch.from_numpy(sides).cuda())
    assert bool((cnt[:, 0] == cn.int()).all())
    root_first = first[:, 0].long()
    k = torch.arange(128, device="cuda")[None, :]
    acts = torch.gather(av["action"][:, :ncols].int() & 0xFFFF, 1, (root_first[:, None] + k).clamp(max=ncols - 1))
    valid = k < cn.int()[:, None]
    assert bool((acts == (mv.int() & 0xFFFF))[valid].all())
    # depth: with 800 simulations some slot must have searched deeper than 3 plies
    assert st["depth_sum"] > 2 * st["sims"]


def test_selfplay_samples_at_full_size():
    import torch
    from xiangqi_alphazero_amd import engine, hip
    cfg = engine.make_config(GAMES, SIMS, max_game_length=400, random_opening_moves=8, temperature_threshold=20,
                             enable_resign=True, resign_threshold=-0.3, resign_check_steps=1, seed=77)   # early resignations
    eng = engine.SelfPlayEngine(cfg, evaluator=PseudoNet("cuda"))
    steps = 12 * (SIMS + 1) + 40                                     # twelve searched plies: resignation needs > 10 samples
    for _ in range(steps):
        eng.step()
    st = eng.stats()
    assert st["overflow"] == 0 and st["samples_dropped"] == 0
    assert st["moves_played"] >= 11 * GAMES
    samples, results = eng.drain()
    assert len(samples) == st["samples_written"] and len(samples) > 1000 and st["resigns"] > 0
    # samples leave the device when their game ends; with 800 simulations per move only resignations and early mates
    # have finished by now -- check whatever came out, and the live trees through the per-slot state instead
    for s in samples[:2000]:
        n = int(s["n_moves"])
        assert int(s["visits"][:n].sum()) == SIMS
    if len(samples):
        b = torch.from_numpy(np.ascontiguousarray(samples["board"])).cuda()
        sd = torch.from_numpy(np.ascontiguousarray(samples["side"]).astype(np.int8)).cuda()
        mv, cn, _, _ = hip.movegen(b, sd)
        assert (cn.cpu().numpy().astype(np.int64) == samples["n_moves"].astype(np.int64)).all()
        k = np.arange(128)[None, :]
        valid = k < samples["n_moves"].astype(np.int64)[:, None]
        assert ((mv.cpu().numpy().view(np.uint16) == samples["actions"])[valid]).all()
    ints = eng.slot_ints.cpu().numpy()
    assert ((ints[:, 1] >= 0) & (ints[:, 1] <= 400)).all()            # plies of the running games
    # finished games: z of every sample is the result seen from the side to move, lengths within the rules' bound
    by_game = {(int(r["slot"]), int(r["game_seq"])): r for r in results}
    assert len(by_game) == len(results) == st["games_finished"]
    for s in samples[:: max(1, len(samples) // 2000)]:
        r = by_game[(int(s["slot"]), int(s["game_seq"]))]
        w = int(r["winner"])
        assert int(s["z"]) == (0 if w == 0 else (1 if w == int(s["side"]) else -1))
        assert int(s["ply"]) < int(r["steps"]) <= 200
    assert ((ints[:, 4] >= 0) & (ints[:, 4] <= SIMS)).all()
    assert set(np.unique(ints[:, 0])).issubset({-1, 1})
