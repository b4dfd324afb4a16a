"""Counter-based HBM-side traffic of the tree-walk kernels ON THE PRODUCT PATH (profiles/rNN_pmc_tree_kernels.json).

    python tools/summarize_pmc_tree.py --bench gpurun_out/x/bench.json FETCH_SIZE=dir WRITE_SIZE=dir [RDREQ=dir] [WRREQ=dir]

Each `NAME=dir` is the output directory of ONE `rocprofv3 --pmc ... --kernel-trace` pass over the same bench command
(counters are collected in separate passes: /opt/skills/guides/MI355X_MICROARCH.md).  Only the k_select / k_expand launches of
REAL steps are summarised -- a step whose k_select is followed by the evaluator's k_stem_conv before the next k_select -- so the
network-free prewarm never enters the averages; the first run of consecutive real steps is the near-uniform variant, the
second the peaked one (bench.py measures them in that order).  Bytes:
  * FETCH_SIZE / WRITE_SIZE: raw counter (KB).  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reads HALF the bytes of 16-B /
    lane streaming reads; for the tree kernels' narrow gathers that correction is uncalibrated, so when the RDREQ pass is
    present the read bytes are taken from the request-size counters instead: 32 n32 + 64 n64 + 128 n128 (n64 = requests that
    are neither 32 B nor 128 B), and write bytes from WRREQ: 64 n64 + 32 (n - n64).
  * `--bench` = the JSON line of the same command WITHOUT a profiler: algorithmic bytes per launch and the HIP-event launch
    times come from there (counters slow the run down), so GB/s = counter bytes / un-profiled launch time.
"""
import argparse
import csv
import glob
import json
import sys
from collections import defaultdict


def load(directory):
    """-> list of (dispatch id, kernel name, {counter: value}, start, end) sorted by dispatch id."""
    per = {}
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                d = int(row["Dispatch_Id"])
                e = per.setdefault(d, [row["Kernel_Name"], defaultdict(float), int(row["Start_Timestamp"]), int(row["End_Timestamp"])])
                e[1][row["Counter_Name"]] += float(row["Counter_Value"])
    return [(d, v[0], v[1], v[2], v[3]) for d, v in sorted(per.items())]


def real_steps(rows):
    """-> [variant 0 launches, variant 1 launches], each {"k_select": [...], "k_expand": [...]} of rows of real steps."""
    steps, cur = [], None
    for r in rows:
        name = r[1]
        if "k_select" in name:
            if cur is not None:
                steps.append(cur)
            cur = {"sel": r, "stem": False, "exp": None}
        elif cur is not None and "k_stem_conv" in name:
            cur["stem"] = True
        elif cur is not None and "k_expand" in name:
            cur["exp"] = r
    if cur is not None:
        steps.append(cur)
    runs, run = [], []
    for s in steps:
        if s["stem"] and s["exp"] is not None:
            run.append(s)
        elif run:
            runs.append(run)
            run = []
    if run:
        runs.append(run)
    return [{"k_select": [s["sel"] for s in r], "k_expand": [s["exp"] for s in r]} for r in runs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", required=True)
    ap.add_argument("passes", nargs="+")
    a = ap.parse_args()
    bench = json.loads([l for l in open(a.bench).read().splitlines() if l.startswith("{")][-1])
    passes = dict(p.split("=", 1) for p in a.passes)
    data = {name: real_steps(load(d)) for name, d in passes.items()}
    variants = [("near-uniform (headline weights)", bench), ("peaked (policy_gain 8)", bench.get("peaked"))]
    out = {"command": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps %d --warmup %d --complete-games 0 --cpu-seconds 0; "
                      "one pass per counter set" % (bench["steps"], bench["warmup"]),
           "selection": "k_select / k_expand launches of real steps only (a k_stem_conv launch between this k_select and the next); "
                        "the network-free prewarm is excluded", "variants": {}}
    for vi, (vname, b) in enumerate(variants):
        if b is None:
            continue
        entry = {}
        for kern in ("k_select", "k_expand"):
            e = {}
            for pname, runs in data.items():
                if vi >= len(runs):
                    continue
                rows = runs[vi][kern]
                n = len(rows)
                c = defaultdict(float)
                for r in rows:
                    for k, v in r[2].items():
                        c[k] += v / n
                e.setdefault("launches_summarised", {})[pname] = n
                e.setdefault("avg_launch_us_under_the_profiler", {})[pname] = round(sum(r[4] - r[3] for r in rows) / n / 1e3, 2)
                for k, v in c.items():
                    e.setdefault("counters_avg_per_launch", {})[k] = round(v, 1)
            c = e.get("counters_avg_per_launch", {})
            rd = wr = None
            if "TCC_EA0_RDREQ_sum" in c:
                n32, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
                rd = 32 * n32 + 128 * n128 + 64 * (c["TCC_EA0_RDREQ_sum"] - n32 - n128)
                e["read_bytes_source"] = "TCC_EA0_RDREQ by request size (32 / 64 / 128 B)"
            elif "FETCH_SIZE" in c:
                rd = c["FETCH_SIZE"] * 1024
                e["read_bytes_source"] = "FETCH_SIZE raw (gfx950: may under-count wide reads by up to 2x)"
            if "TCC_EA0_WRREQ_sum" in c:
                n64 = c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
                wr = 64 * n64 + 32 * (c["TCC_EA0_WRREQ_sum"] - n64)
                e["write_bytes_source"] = "TCC_EA0_WRREQ by request size (32 / 64 B)"
            elif "WRITE_SIZE" in c:
                wr = c["WRITE_SIZE"] * 1024
                e["write_bytes_source"] = "WRITE_SIZE raw"
            tr = b["tree_roofline"][kern]
            ms = b["breakdown_ms"]["select" if kern == "k_select" else "expand_backup"]
            alg = tr["achieved"] * 1e9 * ms * 1e-3                            # algorithmic bytes per launch
            e["algorithmic_bytes_per_launch"] = int(alg)
            e["launch_ms_unprofiled_hip_events"] = ms
            if rd is not None and wr is not None:
                e["hbm_side_bytes_per_launch"] = {"read": int(rd), "write": int(wr), "total": int(rd + wr)}
                e["counter_GBps_at_unprofiled_launch_time"] = round((rd + wr) / (ms * 1e-3) / 1e9, 1)
                e["frac_of_8TBps"] = round((rd + wr) / (ms * 1e-3) / 1e9 / 8000.0, 4)
                e["traffic_over_algorithmic"] = round((rd + wr) / max(alg, 1.0), 3)
            entry[kern] = e
        entry["tree"] = b["tree"]
        out["variants"][vname] = entry
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
