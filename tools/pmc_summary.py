"""Summarise rocprofv3 CSV output: per-kernel averages of kernel-trace durations and PMC counters.

    python tools/pmc_summary.py gpurun_out/prof_bench gpurun_out/prof_pmc1 ...
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("vitk::", "").replace("void ", "")
    return name[:70]


def main(dirs):
    for d in dirs:
        for path in sorted(glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)):
            base = os.path.basename(path)
            with open(path) as f:
                rows = list(csv.DictReader(f))
            if not rows:
                continue
            if base.endswith("kernel_stats.csv"):
                print(f"== {path}")
                for r in rows[:14]:
                    print(f"  {short(r['Name']):70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:10.1f} "
                          f"total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} pct={r['Percentage']}")
            elif base.endswith("counter_collection.csv"):
                print(f"== {path}")
                acc = defaultdict(lambda: defaultdict(list))
                for r in rows:
                    acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, cs in acc.items():
                    line = "  " + f"{k:70s}"
                    for c, v in sorted(cs.items()):
                        line += f" {c}={sum(v) / len(v):.4g}"
                    print(line)


if __name__ == "__main__":
    main(sys.argv[1:])
