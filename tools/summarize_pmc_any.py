"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories (profiles/*_pmc_sq_tcc_*.json).

    python tools/summarize_pmc_any.py <kernel-substring> dir1 [dir2 ...] > out.json
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    want = sys.argv[1]
    per = defaultdict(lambda: defaultdict(float))               # counter -> dispatch -> value
    for d in sys.argv[2:]:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if want in row["Kernel_Name"]:
                        per[row["Counter_Name"]][(d, row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = {k: {"avg_per_launch": sum(v.values()) / len(v), "launches": len(v)} for k, v in sorted(per.items())}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
