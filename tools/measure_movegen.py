#!/usr/bin/env python3
"""Move-generation throughput: positions/s of xq_movegen_batch on the perft(4)
frontier of the opening (3 290 240 positions), beside the C oracle (oracle/xq_oracle.c through ctypes) on one host core.
The reference's own Cython engine is timed in the BUILD container only (tools/calibrate_cpu_port.py); nothing derived
from the reference runs on the GPU box from this tool."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    from oracle import xq_oracle as O
    from xiangqi_alphazero_amd import hip

    boards = torch.from_numpy(O.initial_board().reshape(1, 90).copy()).cuda()
    side = torch.ones(1, dtype=torch.int8, device="cuda")
    for _ in range(4):                                   # frontier after 4 plies
        moves, counts, _, _ = hip.movegen(boards, side)
        cnt = counts.to(torch.int64) & 0xFFFF
        parent = torch.repeat_interleave(torch.arange(boards.shape[0], device="cuda", dtype=torch.int32), cnt)
        mask = torch.arange(128, device="cuda").unsqueeze(0) < cnt.unsqueeze(1)
        boards, side = hip.apply_moves(boards, side, parent.contiguous(), moves[mask].contiguous())
    n = boards.shape[0]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hip.movegen(boards, side)
    e0.record()
    reps = 5
    for _ in range(reps):
        moves, counts, chk, status = hip.movegen(boards, side)
    e1.record()
    torch.cuda.synchronize()
    gpu_s = e0.elapsed_time(e1) / 1e3 / reps
    total_moves = int((counts.to(torch.int64) & 0xFFFF).sum().item())
    hb = boards[:200000].cpu().numpy()
    hs = side[:200000].cpu().numpy()
    t0 = time.perf_counter()
    acc = 0
    for i in range(len(hb)):
        acc += len(O.legal_actions(hb[i], int(hs[i])))
    cpu_s = (time.perf_counter() - t0) / len(hb)
    out = {"positions": n, "perft5": total_moves, "gpu_ms_per_launch": round(gpu_s * 1e3, 3),
           "gpu_positions_per_s": round(n / gpu_s), "gpu_ns_per_position": round(gpu_s / n * 1e9, 2),
           "oracle_c_us_per_position_1core_via_ctypes": round(cpu_s * 1e6, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
