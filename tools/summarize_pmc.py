"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel averages (profiles/*_pmc_hbm_traffic.json).

    python tools/summarize_pmc.py FETCH_SIZE=gpurun_out/pmc_fetch WRITE_SIZE=gpurun_out/pmc_write > profiles/rNN_pmc_hbm_traffic.json

Each argument names a counter and the rocprofv3 output directory of its own pass (counters are collected in separate
passes, /opt/skills/guides/MI355X_MICROARCH.md).  Values are the raw counter sums per launch in KB; on gfx950
FETCH_SIZE counts 16-byte-per-lane reads at half their size, which bench.py corrects (x2) when it reads this file.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def summarise(directory, counter):
    per = defaultdict(lambda: defaultdict(float))           # kernel -> dispatch id -> value
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                per[row["Kernel_Name"][:100]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {"launches": len(v), "avg_KB_per_launch_raw": round(sum(v.values()) / len(v), 1)} for k, v in per.items()}


def main():
    out = {}
    for arg in sys.argv[1:]:
        counter, directory = arg.split("=", 1)
        out[counter] = summarise(directory, counter)
    out["note"] = ("rocprofv3 --pmc <counter> --kernel-trace, one pass per counter, over the bench command of tools/run_pmc_passes.sh "
                   "(`python3 bench.py --steps 4 --warmup 2 --prewarm 300 --complete-games 0 --cpu-seconds 0`); raw counter units (KB); "
                   "FETCH_SIZE needs the gfx950 x2 correction for 16-B/lane reads")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
