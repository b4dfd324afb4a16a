#!/usr/bin/env python3
"""Train-step throughput: samples/s of `training.train_network` on the GPU -- device-resident
compact replay buffer, batches materialised by xq_samples_to_batch, torch autograd for forward/backward/Adam -- beside
the reference's step restated on the host cores (train.py:376-447: dense (state, pi, z) tuples through a DataLoader,
same module, same loss, same optimiser; `--cpu-threads` torch threads).  Synthetic samples from a short self-play run.

    python tools/measure_train_step.py [--channels 128 --blocks 6 --samples 4096 --batch 256 --cpu-batches 4]
"""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--samples", type=int, default=4096, help="logical samples in the buffer (records x 2 mirrors)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--cpu-batches", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--fused-adam", type=int, default=1, help="torch.optim.Adam(fused=True), as train_loop.AlphaZeroLoop builds it on the GPU")
    ap.add_argument("--native-conv", type=int, default=1, help="1: tower convolutions (forward + data gradient) on the hand-written "
                    "Winograd kernel (XiangqiNet.use_native_conv); 0: torch autograd on the ROCm library throughout")
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.nn.functional as F
    from xiangqi_alphazero_amd import model, selfplay, training, weights
    from xiangqi_alphazero_amd.sample_format import to_reference_tuples

    cfg = types.SimpleNamespace(num_simulations=16, c_puct=1.5, temperature_threshold=20, max_game_length=120,
                                random_opening_moves=8, enable_resign=False, resign_threshold=-0.9, resign_check_steps=5)
    gen = model.XiangqiNet(64, 3)
    gen.load_state_dict(weights.make_state_dict(64, 3))
    samples, results, _, _ = selfplay.run_games(gen, cfg, 64, "cuda", seed=5)
    samples = samples[:a.samples // 2]
    buf = training.ReplayBuffer(10 ** 6, "cuda")
    buf.extend(samples)
    net = model.XiangqiNet(a.channels, a.blocks)
    net.load_state_dict(weights.make_state_dict(a.channels, a.blocks))
    net = net.cuda()
    net.use_native_conv(bool(a.native_conv))
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-4, fused=bool(a.fused_adam))   # as AlphaZeroLoop builds it
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50, 80], gamma=0.1)
    tcfg = types.SimpleNamespace(min_buffer_size=1, num_epochs=1, batch_size=a.batch)
    training.train_network(net, opt, sch, buf, tcfg)                      # warm-up epoch (allocator, MIOpen find)
    torch.cuda.synchronize()
    tcfg.num_epochs = a.epochs
    t0 = time.perf_counter()
    stats = training.train_network(net, opt, sch, buf, tcfg)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    gpu_rate = a.epochs * len(buf) / gpu_s

    # the reference's step on the host: dense tuples, DataLoader-style batches, same math (train.py:398-419)
    from oracle.cpu_baseline import host_cores                         # CPUs this job is actually granted (cgroup quota)
    torch.set_num_threads(a.cpu_threads if a.cpu_threads > 0 else host_cores())
    dense, _ = to_reference_tuples(samples[:a.cpu_batches * a.batch // 2 + 1], results, augment=True)
    dense = dense[:a.cpu_batches * a.batch]
    cpu_net = model.XiangqiNet(a.channels, a.blocks)
    cpu_net.load_state_dict(weights.make_state_dict(a.channels, a.blocks))
    cpu_net.train()
    copt = torch.optim.Adam(cpu_net.parameters(), lr=2e-3, weight_decay=1e-4)
    t0 = time.perf_counter()
    done = 0
    for lo in range(0, len(dense), a.batch):
        chunk = dense[lo:lo + a.batch]
        states = torch.FloatTensor(np.stack([c[0] for c in chunk]))
        pis = torch.FloatTensor(np.stack([c[1] for c in chunk]))
        zs = torch.FloatTensor(np.array([[c[2]] for c in chunk]))
        logits, value = cpu_net(states)
        loss = -torch.mean(torch.sum(pis * F.log_softmax(logits, dim=1), dim=1)) + F.mse_loss(value, zs)
        copt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(cpu_net.parameters(), 1.0)
        copt.step()
        done += len(chunk)
    cpu_s = time.perf_counter() - t0
    print(json.dumps({
        "net": "%dx%d" % (a.channels, a.blocks), "batch": a.batch, "buffer_samples": len(buf), "epochs_timed": a.epochs,
        "gpu_samples_per_s": round(gpu_rate, 1), "gpu_ms_per_batch": round(1e3 * gpu_s / (a.epochs * -(-len(buf) // a.batch)), 2),
        "gpu_path": "device-resident compact buffer + xq_samples_to_batch + torch autograd (fp32); tower convolutions: "
                    + ("forward and data gradient on xq_wino_conv3x3 (channels-last), weight gradient ROCm library" if a.native_conv
                       else "ROCm library"),
        "native_conv": bool(a.native_conv), "fused_adam": bool(a.fused_adam),
        "cpu_samples_per_s": round(done / cpu_s, 1), "cpu_threads": torch.get_num_threads(), "cpu_samples_timed": done,
        "cpu_path": "the reference's train step restated (train.py:398-419), dense tuples, torch CPU",
        "policy_loss": stats.get("policy_loss")}))


if __name__ == "__main__":
    main()
