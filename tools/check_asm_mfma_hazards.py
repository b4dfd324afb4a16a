#!/usr/bin/env python3
"""Static check of the inline-asm MFMAs in csrc/*.hip (the accumulator tiles pinned to VGPRs: xq_conv.hip, xq_conv_bf16.hip, xq_train.hip).

The compiler's hazard recogniser pads ITS OWN MFMAs (`s_nop`) when a vector instruction writes one of their source registers less than
two wait states earlier; an MFMA inside `asm volatile` is opaque to it.  Round 3 met that hazard twice (a deterministic wrong tile in
k_wino_wgrad; a v_cvt_pk_bf16_f32 in front of k_wino_conv_bf16's asm MFMA).  This tool compiles every kernel file to gfx950 assembly and
fails when an asm MFMA that does not carry its own `s_nop` has a vector write of one of its A/B source registers among the two
instructions in front of it.  tests/test_host_logic.py runs it, so a compiler or scheduling change cannot reintroduce the hazard silently.

    python tools/check_asm_mfma_hazards.py        # exit 0 = clean
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xiangqi-alphazero_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
         "-S", "--cuda-device-only"]


def regs(tok: str):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm_path: str):
    """-> (number of inline-asm MFMAs, [description of each suspicious one])"""
    lines = [l.strip() for l in open(asm_path)]
    total, bad = 0, []
    for i, l in enumerate(lines):
        if not l.startswith(";;#ASMSTART"):
            continue
        body = []
        j = i + 1
        while j < len(lines) and not lines[j].startswith(";;#ASMEND"):
            body.append(lines[j]); j += 1
        mf = [b for b in body if b.startswith("v_mfma")]
        if not mf:
            continue
        total += 1
        if any(b.startswith("s_nop") for b in body):
            continue                                              # carries its own wait states
        ops = mf[0].split(None, 1)[1].split(",")
        src = regs(ops[1]) | regs(ops[2])
        seen, k = 0, i - 1
        while k >= 0 and seen < 2:
            p = lines[k]; k -= 1
            if not p or p.startswith(";") or p.startswith(".") or p.endswith(":"):
                continue
            seen += 1
            if p.startswith("s_nop"):
                break
            if p.startswith("v_") and not p.startswith("v_mfma") and regs(p.split(None, 1)[1].split(",")[0]) & src:
                bad.append("%s: `%s` right before `%s`" % (os.path.basename(asm_path), p, mf[0]))
                break
    return total, bad


def main() -> int:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    failures, report = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            if "v_mfma" not in open(src).read():
                continue
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            subprocess.run([hipcc] + FLAGS + [src, "-o", out], check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            total, bad = scan(out)
            report.append("%s: %d inline-asm MFMAs, %d suspicious" % (os.path.basename(src), total, len(bad)))
            failures += bad
    print("\n".join(report + failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
