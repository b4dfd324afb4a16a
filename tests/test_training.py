"""Next rows: train step on the device-resident compact buffer, checkpoint round trip."""
import json
import os

import numpy as np
import pytest

import golden_io as G
from test_host_logic import _oracle_game_as_compact

REF = "/root/reference/training"


def _trace(name="train_trace.json"):
    return json.load(open(os.path.join(G.GOLDEN, name)))


@pytest.mark.gpu
def test_batch_materialisation_matches_dense_adapter():
    """xq_samples_to_batch == the reference-schema adapter (planes, dense pi incl. mirror, z) for every logical sample."""
    import torch
    from xiangqi_alphazero_amd import training
    from xiangqi_alphazero_amd.sample_format import to_reference_tuples
    for t in G.game_traces()[:2]:
        arr, res = _oracle_game_as_compact(t)
        data, _ = to_reference_tuples(arr, res, augment=True)
        buf = training.ReplayBuffer(50000)
        buf.extend(arr)
        assert len(buf) == len(data)
        states, pi, z = buf.batch(torch.arange(len(data)))
        states, pi, z = states.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
        for j, (s, p, zz) in enumerate(data):
            np.testing.assert_array_equal(states[j], s)
            assert z[j, 0] == zz
            assert np.array_equal(np.nonzero(pi[j])[0], np.nonzero(p)[0])
            np.testing.assert_allclose(pi[j], p.astype(np.float32), rtol=2e-7, atol=0)


@pytest.mark.gpu
def test_replay_buffer_is_fifo_like_the_reference_deque():
    import torch
    from xiangqi_alphazero_amd import training
    arr, _ = _oracle_game_as_compact(G.game_traces()[1])
    buf = training.ReplayBuffer(max_size=20)            # 10 records
    buf.extend(arr[:7]); buf.extend(arr[7:16])
    assert len(buf) == 20
    st, _, _ = buf.batch(torch.tensor([0, 18]))
    from xiangqi_alphazero_amd.sample_format import encode_planes
    np.testing.assert_array_equal(st[0].cpu().numpy(), encode_planes(arr[6]["board"], int(arr[6]["side"])))
    np.testing.assert_array_equal(st[1].cpu().numpy(), encode_planes(arr[15]["board"], int(arr[15]["side"])))


@pytest.mark.gpu
@pytest.mark.parametrize("channels,co_block", [(64, 64), (128, 64), (128, 128), (256, 128), (512, 64)])
def test_device_filter_transform_equals_the_host_transform(channels, co_block):
    """xq_wino_transform_filters (one launch on the device, what the train step uses every optimizer step) against the host's
    float64 einsum (hip.wino_transform_weights, what self-play uses once per weight update): the same float32 values; the
    data-gradient form equals the host transform of the transposed, 180-degree-rotated filters."""
    import torch
    from xiangqi_alphazero_amd import hip
    w = (torch.randn(channels, channels, 3, 3, generator=torch.Generator().manual_seed(channels + co_block)) * 0.05).cuda()
    for dgrad in (False, True):
        host = hip.wino_transform_weights(w.flip(2, 3).transpose(0, 1).contiguous() if dgrad else w, co_block)
        dev = hip.wino_transform_filters_device(w, co_block, dgrad)
        assert dev.shape == host.shape
        # float64 sums of three products taken in a different order may round differently in the last float32 bit
        assert (dev - host).abs().max().item() <= 1.2e-7 * host.abs().max().item()
        assert (dev != host).double().mean().item() < 1e-3
    both = hip.wino_transform_filters_device(w, co_block, both=True)                 # one launch: [0] forward, [1] data gradient
    assert both.shape == (2,) + host.shape
    assert torch.equal(both[0], hip.wino_transform_filters_device(w, co_block, False))
    assert torch.equal(both[1], hip.wino_transform_filters_device(w, co_block, True))


@pytest.mark.gpu
@pytest.mark.parametrize("channels,batch", [(64, 5), (64, 1), (128, 37), (256, 256), (512, 3)])
def test_native_conv_forward_and_gradients_equal_torch(channels, batch):
    """native_conv.WinoConv3x3 (forward, data gradient and weight gradient on the hand-written kernels) against torch's convolution in
    float64 on the same operands: output, dL/dx and dL/dw within 1e-5 of the largest magnitude (float32 Winograd rounding)."""
    import torch
    import torch.nn.functional as F
    from xiangqi_alphazero_amd import native_conv
    gen = torch.Generator().manual_seed(batch)
    x = torch.randn(batch, channels, 10, 9, generator=gen).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(channels, channels, 3, 3, generator=gen) * (2.0 / (9 * channels)) ** 0.5).cuda().requires_grad_(True)
    gy = torch.randn(batch, channels, 10, 9, generator=gen).cuda()
    y = native_conv.conv3x3(x, w)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, padding=1)
    y64.backward(gy.double())
    for got, want in ((y, y64), (x.grad, x64.grad), (w.grad, w64.grad)):
        assert (got.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    x2 = x.detach().contiguous().requires_grad_(True)                        # NCHW-contiguous input: same values (one copy inside)
    assert torch.equal(native_conv.conv3x3(x2, w), y)


@pytest.mark.gpu
@pytest.mark.parametrize("channels,batch,with_res,relu", [(64, 3, False, True), (128, 20, True, True), (256, 256, True, True),
                                                          (256, 64, False, False), (512, 7, True, False)])
def test_fused_batchnorm_train_equals_torch_float64(channels, batch, with_res, relu):
    """native_conv.BnAct (training-mode BatchNorm2d + skip-add + ReLU, hand-written) against torch in float64 on the same operands:
    output, running statistics, and the gradients of x, the skip input, gamma and beta."""
    import torch
    import torch.nn.functional as F
    from xiangqi_alphazero_amd import native_conv
    gen = torch.Generator().manual_seed(channels + batch)
    mk = lambda *shape: torch.randn(*shape, generator=gen)
    x = (mk(batch, channels, 10, 9) * 1.7 + 0.3).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = mk(batch, channels, 10, 9).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    bn = torch.nn.BatchNorm2d(channels).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(mk(channels).abs() + 0.5); bn.bias.copy_(mk(channels) * 0.2)
        bn.running_mean.copy_(mk(channels) * 0.1); bn.running_var.copy_(mk(channels).abs() + 0.5)
    rm0, rv0 = bn.running_mean.double().clone(), bn.running_var.double().clone()
    gy = mk(batch, channels, 10, 9).cuda()
    assert native_conv.bn_supported(bn)
    y = native_conv.bn_act(x, bn, r, relu)
    y.backward(gy)
    assert int(bn.num_batches_tracked) == 1
    x64 = x.detach().double().requires_grad_(True)
    r64 = r.detach().double().requires_grad_(True) if with_res else None
    g64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    y64 = F.batch_norm(x64, rm0, rv0, g64, b64, True, bn.momentum, bn.eps)
    if with_res:
        y64 = y64 + r64
    if relu:
        y64 = F.relu(y64)
    y64.backward(gy.double())
    pairs = [(y, y64), (x.grad, x64.grad), (bn.weight.grad, g64.grad), (bn.bias.grad, b64.grad), (bn.running_mean, rm0), (bn.running_var, rv0)]
    if with_res:
        pairs.append((r.grad, r64.grad))
    for i, (got, want) in enumerate(pairs):
        assert (got.double() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), i
    bn.eval()
    assert not native_conv.bn_supported(bn)


@pytest.mark.gpu
@pytest.mark.parametrize("trace,native", [("train_trace.json", False), ("train_trace_64x2.json", False), ("train_trace_64x2.json", True)])
def test_train_network_matches_reference_trace(trace, native):
    """Same data, weights, optimiser and batch order as the reference's train_network run recorded in the fixture -- with the
    tower's convolutions on the ROCm library and (native) on the hand-written Winograd kernel (XiangqiNet.use_native_conv):
    losses within 1e-4 relative; probed weights within 1e-4 absolute -- they move by up to 1.2e-2 in the six Adam steps
    (lr 2e-3) and Adam's normalised update amplifies last-bit gradient differences between GPU and CPU kernels."""
    import types
    import torch
    from xiangqi_alphazero_amd import model, training, weights
    t = _trace(trace)
    game = [x for x in G.game_traces() if x["name"] == t["game"]][0]
    arr, _ = _oracle_game_as_compact(game)
    buf = training.ReplayBuffer(50000)
    buf.extend(arr)
    assert len(buf) == t["n_samples"]
    net = model.XiangqiNet(*t["net"])
    net.load_state_dict(weights.make_state_dict(*t["net"], seed=t["seed"]))
    net = net.cuda()
    if native:
        net.use_native_conv(True)
    opt = torch.optim.Adam(net.parameters(), lr=t["lr"], weight_decay=t["weight_decay"])
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=t["milestones"], gamma=t["gamma"])
    cfg = types.SimpleNamespace(min_buffer_size=10, num_epochs=t["num_epochs"], batch_size=t["batch_size"])
    stats = training.train_network(net, opt, sch, buf, cfg, shuffle=False)
    # 64x2: a float32 forward differs from the reference's CPU run in the last bits, which flips the ReLU mask of the few activations
    # that sit within ~1e-6 of zero; each flip changes gradient entries by their full magnitude and six Adam steps (lr 2e-3) amplify
    # it: the ROCm library path itself lands 4e-3 from the recorded value loss here (and the hand-written path 1e-6 .. 3e-4 from the
    # library path; value_head.1.running_var ends 1.3 % off for BOTH).  Both are held to 1e-2 on the losses and 3e-2 relative + 2e-3
    # absolute on the probed weights at this size; the tight bound stays on the 16x1 trace.
    rtol, atol, prtol = (1e-4, 1e-4, 0.0) if t["net"][0] == 16 else (1e-2, 2e-3, 3e-2)
    for k in ("policy_loss", "value_loss", "total_loss", "learning_rate"):
        assert abs(stats[k] - t["stats"][k]) <= rtol * abs(t["stats"][k]) + 1e-7, (k, stats[k], t["stats"][k])
    sd = net.state_dict()
    assert int(sd["input_conv.1.num_batches_tracked"]) == t["num_batches_tracked"]
    for k, want in t["probe"].items():
        np.testing.assert_allclose(sd[k].flatten()[:8].double().cpu().numpy(), want, rtol=prtol, atol=atol, err_msg=k)
    assert training.train_network(net, opt, sch, training.ReplayBuffer(100), cfg) == {}     # below min_buffer_size


@pytest.mark.gpu
def test_native_step_gradients_against_float64():
    """One forward/backward of the whole training module (train mode, batch statistics) with XiangqiNet.use_native_conv -- hand-written
    convolutions (forward, data and weight gradient) and fused BatchNorm -- against the same module in float64 on the CPU: loss within
    1e-6; every parameter gradient within 3e-2 of its float64 value in the L2 norm.  The bound is a norm and not 1e-5 per entry because
    a float32 forward flips the ReLU mask of the few pre-activations within ~1e-6 of zero, and one flip moves gradient entries by their
    full size (the ROCm library step shows the same 1e-3 .. 4e-2 per-entry deviations, tests/microbench/train_grad_check.py); each
    kernel by itself is held to 1e-5 / 2e-6 per entry in the tests above, also on the tensors of this very step
    (tests/microbench/train_grad_detail.py: 4e-7 .. 2e-6)."""
    import copy
    import torch
    import torch.nn.functional as F
    from xiangqi_alphazero_amd import model, weights
    net = model.XiangqiNet(128, 3)
    net.load_state_dict(weights.make_state_dict(128, 3, seed=9))
    net.train()
    gen = torch.Generator().manual_seed(4)
    x = (torch.rand(96, 15, 10, 9, generator=gen) < 0.1).float()
    pi = torch.softmax(torch.randn(96, 8100, generator=gen), 1)
    z = torch.rand(96, 1, generator=gen) * 2 - 1

    def run(m, dev, dt):
        logits, value = m(x.to(dev, dt))
        loss = -torch.mean(torch.sum(pi.to(dev, dt) * F.log_softmax(logits, dim=1), dim=1)) + F.mse_loss(value, z.to(dev, dt))
        loss.backward()
        return loss.item(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}

    l64, g64 = run(copy.deepcopy(net).double(), "cpu", torch.float64)
    nat = copy.deepcopy(net).cuda().use_native_conv(True)
    assert nat.res_blocks[0].native_conv and list(nat.state_dict().keys()) == list(net.state_dict().keys())
    assert nat.res_blocks[0].conv1.weight.is_contiguous()               # tower filters stay plain [C, C, 3, 3]
    l32, g32 = run(nat, "cuda", torch.float32)
    assert abs(l32 - l64) <= 1e-6 * abs(l64)
    for name, want in g64.items():
        assert (g32[name] - want).norm().item() <= 3e-2 * want.norm().item() + 1e-12, name


def test_checkpoint_files_have_the_reference_layout(tmp_path):
    import torch
    from xiangqi_alphazero_amd import model, training, weights
    cur, best = model.XiangqiNet(16, 1), model.XiangqiNet(16, 1)
    cur.load_state_dict(weights.make_state_dict(16, 1, seed=1)); best.load_state_dict(weights.make_state_dict(16, 1, seed=2))
    opt = torch.optim.Adam(cur.parameters(), lr=0.002, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50, 80], gamma=0.1)
    path = training.save_checkpoint(str(tmp_path), 7, cur, best, opt, sch, total_games=123, is_best=True)
    ck = torch.load(path, weights_only=True)
    assert set(ck) == {"iteration", "model_state_dict", "best_model_state_dict", "optimizer_state_dict",
                       "scheduler_state_dict", "config", "total_games"}                      # train.py:539-550
    bm = torch.load(os.path.join(tmp_path, "best_model.pt"), weights_only=True)
    assert set(bm) == {"model_state_dict", "config", "iteration", "total_games"} and bm["config"] == {"num_channels": 16, "num_res_blocks": 1}
    c2, b2 = model.XiangqiNet(16, 1), model.XiangqiNet(16, 1)
    info = training.load_checkpoint(path, c2, b2, torch.optim.Adam(c2.parameters()), None)
    assert info["iteration"] == 7 and info["total_games"] == 123
    assert all(torch.equal(a, b) for a, b in zip(c2.state_dict().values(), cur.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(b2.state_dict().values(), best.state_dict().values()))
    assert torch.equal(model.load_reference_checkpoint(os.path.join(tmp_path, "best_model.pt")).state_dict()["policy_head.4.bias"],
                       best.state_dict()["policy_head.4.bias"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_checkpoint_loads_with_the_reference_loader(tmp_path):
    """A file written here is read by the reference's own AlphaZeroTrainer.load_checkpoint (train.py:569-579)."""
    import sys
    import types
    import torch
    from xiangqi_alphazero_amd import model, training, weights
    sys.dont_write_bytecode = True
    cur, best = model.XiangqiNet(16, 1), model.XiangqiNet(16, 1)
    cur.load_state_dict(weights.make_state_dict(16, 1, seed=5)); best.load_state_dict(weights.make_state_dict(16, 1, seed=6))
    opt = torch.optim.Adam(cur.parameters(), lr=0.002, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50, 80], gamma=0.1)
    path = training.save_checkpoint(str(tmp_path), 3, cur, best, opt, sch, total_games=40)
    sys.path[:0] = [REF, os.path.join(G.GOLDEN, "..", "..", "oracle", "_ref")]
    cwd = os.getcwd()
    os.chdir(str(tmp_path))                          # the reference opens 'training.log' in the cwd at import
    try:
        import train as ref_train
        import model as ref_model
    finally:
        os.chdir(cwd)
    rc, rb = ref_model.XiangqiNet(num_channels=16, num_res_blocks=1), ref_model.XiangqiNet(num_channels=16, num_res_blocks=1)
    ro = torch.optim.Adam(rc.parameters(), lr=0.002, weight_decay=1e-4)
    fake = types.SimpleNamespace(device="cpu", current_model=rc, best_model=rb, optimizer=ro,
                                 scheduler=torch.optim.lr_scheduler.MultiStepLR(ro, milestones=[50, 80], gamma=0.1))
    ref_train.AlphaZeroTrainer.load_checkpoint(fake, path)
    assert fake.iteration == 3 and fake.total_games == 40
    assert all(torch.equal(a, b) for a, b in zip(rc.state_dict().values(), cur.state_dict().values()))


@pytest.mark.gpu
def test_full_loop_two_iterations(tmp_path):
    """AlphaZeroTrainer.train() order on the engine: self-play -> train -> arena gate (iteration 2) -> checkpoint."""
    import types
    import torch
    from xiangqi_alphazero_amd import train_loop
    cfg = types.SimpleNamespace(
        num_channels=64, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=16,
        max_game_length=30, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True, random_opening_moves=4,
        num_iterations=2, batch_size=64, num_epochs=1, learning_rate=0.002, weight_decay=1e-4, lr_milestones=[50, 80],
        lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=100, eval_games=4, eval_win_rate=0.55, eval_simulations=8,
        checkpoint_dir=str(tmp_path), save_interval=2)
    loop = train_loop.AlphaZeroLoop(cfg, "cuda", seed=3)
    stats = loop.train()
    assert [s["iteration"] for s in stats] == [1, 2]
    assert stats[0]["self_play"]["games"] == 16 and stats[0]["self_play"]["mode"] == "hip"
    assert stats[0]["training"]["policy_loss"] > 0 and stats[1]["training"]["policy_loss"] < stats[0]["training"]["policy_loss"] + 1.0
    assert stats[0]["evaluation"] == {} and set(stats[1]["evaluation"]) == {"new_wins", "old_wins", "draws", "win_rate", "model_updated"}
    assert os.path.exists(tmp_path / "checkpoint_iter2.pt") and os.path.exists(tmp_path / "best_model.pt")
    saved = json.load(open(tmp_path / "training_stats.json"))
    assert len(saved) == 2 and saved[1]["self_play"]["buffer_size"] == len(loop.buffer)
    ev = stats[1]["evaluation"]
    same = all(torch.equal(a, b) for a, b in zip(loop.current_model.state_dict().values(), loop.best_model.state_dict().values()))
    assert same                                        # promoted or reverted: both leave current == best (train.py:525-533)
    assert ev["new_wins"] + ev["old_wins"] + ev["draws"] == 4


def _loop_rank_gpu(rank, world, port, tmp):
    import hashlib
    import types
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xiangqi_alphazero_amd import train_loop
    cfg = types.SimpleNamespace(
        num_channels=64, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=12,
        max_game_length=30, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True, random_opening_moves=4,
        num_iterations=2, batch_size=64, num_epochs=1, learning_rate=0.002, weight_decay=1e-4, lr_milestones=[50, 80],
        lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=100, eval_games=5, eval_win_rate=0.55, eval_simulations=8,
        checkpoint_dir=os.path.join(tmp, "ck%d" % rank), save_interval=2)
    loop = train_loop.AlphaZeroLoop(cfg, "cuda", seed=3)
    stats = loop.train()
    flat = torch.cat([t.reshape(-1).double() for t in list(loop.current_model.state_dict().values())
                      + list(loop.best_model.state_dict().values())]).cpu().numpy()
    digest = hashlib.sha256(flat.tobytes() + loop.buffer.store.cpu().numpy().tobytes()).hexdigest()
    ev = stats[1]["evaluation"]
    ok = (stats[0]["self_play"]["games"] == 12 and stats[0]["self_play"]["num_workers"] == 2 and len(loop.buffer) > 100
          and stats[1]["self_play"]["buffer_size"] == len(loop.buffer) and ev["new_wins"] + ev["old_wins"] + ev["draws"] == 5
          and (rank != 0 or stats[0]["training"]["policy_loss"] > 0))
    open(os.path.join(tmp, "gloop%d" % rank), "w").write("%d %s %d" % (int(bool(ok)), digest, len(loop.buffer)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_full_loop_two_ranks_share_one_gpu(tmp_path):
    """BASELINE configs[4]'s loop with world_size 2 (gloo; both ranks on the one GPU of the test box): games and arena
    games sharded, device-resident samples all-gathered, rank-0 training + broadcast.  Both ranks must end with
    bit-identical weights (current and best) and replay buffers."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_loop_rank_gpu, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [open(tmp_path / ("gloop%d" % r)).read().split() for r in range(2)]
    assert out[0][0] == "1" and out[1][0] == "1", out
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


def _ddp_rank(rank, world, port, tmp):
    import types
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xiangqi_alphazero_amd import model, training, weights
    t = _trace()
    game = [x for x in G.game_traces() if x["name"] == t["game"]][0]
    arr, _ = _oracle_game_as_compact(game)
    buf = training.ReplayBuffer(50000)
    buf.extend(arr)
    net = model.XiangqiNet(*t["net"])
    net.load_state_dict(weights.make_state_dict(*t["net"], seed=t["seed"]))
    net = net.cuda()
    opt = torch.optim.Adam(net.parameters(), lr=t["lr"], weight_decay=t["weight_decay"])
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=t["milestones"], gamma=t["gamma"])
    cfg = types.SimpleNamespace(min_buffer_size=10, num_epochs=t["num_epochs"], batch_size=t["batch_size"])
    stats = training.train_network(net, opt, sch, buf, cfg, shuffle=False, ddp=True)
    sd = net.state_dict()
    out = {"stats": stats, "probe": {k: [float(x) for x in sd[k].flatten()[:8].double().cpu()] for k in t["probe"]},
           "nbt": int(sd["input_conv.1.num_batches_tracked"]), "keys": list(sd.keys())}
    json.dump(out, open(os.path.join(tmp, "ddp%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_train_step_matches_reference_trace(tmp_path):
    """train.py:376-447 computed DATA-PARALLEL (world_size 2 over gloo, both ranks on the test box's one GPU): every batch
    split across the ranks, SyncBatchNorm, DDP gradient all-reduce.  Against the SAME recorded run of the reference's
    single-device train_network as the world-1 test: losses within 1e-4 relative, probed weights within 1e-4; both ranks
    end with the same weights; the state_dict keeps the reference's keys."""
    import socket
    import torch.multiprocessing as mp
    from xiangqi_alphazero_amd import weights
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t = _trace()
    outs = [json.load(open(tmp_path / ("ddp%d.json" % r))) for r in range(2)]
    for o in outs:
        for k in ("policy_loss", "value_loss", "total_loss", "learning_rate"):
            assert abs(o["stats"][k] - t["stats"][k]) <= 1e-4 * abs(t["stats"][k]) + 1e-7, (k, o["stats"][k], t["stats"][k])
        assert o["nbt"] == t["num_batches_tracked"]
        for k, want in t["probe"].items():
            np.testing.assert_allclose(o["probe"][k], want, rtol=0, atol=1e-4, err_msg=k)
        assert o["keys"] == list(weights.state_dict_shapes(*t["net"]).keys())
    assert outs[0]["probe"] == outs[1]["probe"]


@pytest.mark.gpu
def test_loop_resumes_on_the_gpu_with_native_step_and_fused_adam(tmp_path):
    """AlphaZeroLoop on the GPU (64 channels: hand-written train step, fused Adam): a loop resumed from the checkpoint of iteration 1
    carries on with iteration 2 -- iteration counter, games, buffer, optimizer state (its step counters included) and both models come
    back, and the resumed model is still routed through the hand-written kernels."""
    import types
    import torch
    from xiangqi_alphazero_amd import train_loop
    cfg = types.SimpleNamespace(
        num_channels=64, num_res_blocks=1, num_simulations=8, c_puct=1.5, temperature_threshold=10, num_games_per_iter=16,
        max_game_length=30, resign_threshold=-0.9, resign_check_steps=5, enable_resign=True, random_opening_moves=4,
        num_iterations=2, batch_size=64, num_epochs=1, learning_rate=0.002, weight_decay=1e-4, lr_milestones=[50, 80],
        lr_gamma=0.1, max_buffer_size=50000, min_buffer_size=100, eval_games=4, eval_win_rate=0.55, eval_simulations=8,
        checkpoint_dir=str(tmp_path), save_interval=1)
    a = train_loop.AlphaZeroLoop(cfg, "cuda", seed=3)
    assert a.current_model.res_blocks[0].native_conv and not a.best_model.res_blocks[0].native_conv
    assert a.optimizer.defaults.get("fused") is True
    a.train(1)
    steps_a = [int(st["step"]) for st in a.optimizer.state.values()]
    assert steps_a and min(steps_a) > 0
    b = train_loop.AlphaZeroLoop(cfg, "cuda", seed=3)
    info = b.resume(str(tmp_path / "checkpoint_iter1.pt"))
    assert info["iteration"] == 1 and info["replay_buffer_restored"] and b.total_games == a.total_games and len(b.buffer) == len(a.buffer)
    assert [int(st["step"]) for st in b.optimizer.state.values()] == steps_a
    for (k, x), y in zip(a.current_model.state_dict().items(), b.current_model.state_dict().values()):
        assert torch.equal(x, y), k
    assert b.current_model.res_blocks[0].native_conv and b.current_model.res_blocks[0].conv1.weight.is_contiguous()
    stats = b.train(2)
    assert [s["iteration"] for s in stats] == [1, 2] and stats[1]["training"]["policy_loss"] > 0
    assert set(stats[1]["evaluation"]) >= {"new_wins", "old_wins", "draws", "win_rate", "model_updated"}
