"""The RCCL code path on the one GPU of the test box: a fresh process initialises
an `nccl` process group of world size 1 before any other GPU work and runs every collective of the multi-GPU path through
it (tests/rccl_world1_worker.py).  Plus, on CPU: `bench.py --gpus 8 --rehearse` starts eight ranks by itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_every_collective_runs_over_rccl_with_one_rank(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")]
    assert line, r.stdout[-2000:]
    o = json.loads(line[-1][len("RCCL_WORLD1 "):])
    assert o["backend"] == "nccl" and o["world"] == 1
    assert o["broadcast_unchanged"] and o["gather_device_ok"] and o["gather_empty_ok"] and o["gather_host_ok"]
    assert o["arena_games"] == 3
    assert o["ddp_uses_syncbn"] and o["ddp_wrapper_cached"] and o["ddp_wrapper_reused"]
    # one rank: the data-parallel step IS the reference's step (same batches, SyncBatchNorm of one rank = BatchNorm):
    # it reproduces the reference's recorded run like the single-device step does (tests/test_training.py: 1e-4 relative)
    for mode in ("ddp", "single"):
        for k in ("policy_loss", "value_loss", "total_loss", "learning_rate"):
            want = o["reference_stats"][k]
            assert abs(o["train_stats"][mode][k] - want) <= 1e-4 * abs(want) + 1e-7, (mode, k, o["train_stats"][mode][k], want)
    assert o["train_stats"]["ddp_again"]["policy_loss"] < o["train_stats"]["ddp"]["policy_loss"] + 1.0
    assert o["loop_iterations"] == [1, 2] and o["loop_games"] == [8, 8] and all(o["loop_trained"])
    assert o["loop_eval_keys"] == ["draws", "model_updated", "new_wins", "old_wins", "win_rate"]
    assert o["loop_grouped"] and o["loop_ddp_wrapper"]


@pytest.mark.gpu
def test_bench_runs_its_collectives_over_rccl_with_one_rank():
    """`XQ_BENCH_FORCE_GROUP=1 python bench.py --gpus 1`: bench.py's own process-group code (init with device_id, barriers
    around the timed region, the all-gather of the per-rank figures on device tensors) over RCCL, tiny workload."""
    env = dict(os.environ, XQ_BENCH_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--games", "256",
                        "--sims", "16", "--channels", "64", "--blocks", "2", "--no-peaked", "--complete-games", "0", "--cpu-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["ranks"]["backend"] == "nccl" and o["ranks"]["world_size_seen"] == 1 and o["n_gpus"] == 1 and o["value"] > 0


def test_bench_starts_eight_ranks_by_itself_rehearsal():
    """`python bench.py --gpus 8 --rehearse` (no GPU API, gloo): the launcher starts eight fresh ranks, they rendezvous and the
    per-rank stand-in figures come back through the same all-gather a real run uses: world_size_seen 8, max elapsed over the
    ranks, min / max per rank.  The line is labelled a rehearsal and carries no value."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    o = json.loads(lines[0])
    assert o["rehearsal"] is True and o["value"] is None and o["n_gpus"] == 8
    rk = o["ranks"]
    assert rk["launched_by"] == "bench.py" and rk["backend"] == "gloo" and rk["world_size_seen"] == 8
    assert rk["elapsed_max_s"] == pytest.approx(1.07) and rk["sims_total"] == 36000.0
    assert rk["sims_per_s_min"] == pytest.approx(1000.0) and rk["sims_per_s_max"] == pytest.approx(8000 / 1.07, rel=1e-3)
