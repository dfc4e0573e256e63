"""Summarise rocprofv3 CSV output: per-kernel averages of kernel-trace durations and PMC counters.

    python tools/pmc_summary.py [--workload NAME] gpurun_out/prof_bench gpurun_out/prof_pmc1 ...

With --workload it also writes traffic_NAME.json next to the first directory: HBM-side bytes per launch of the
named kernels (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; FETCH_SIZE counts 128-byte requests as 64 bytes on
gfx950), the table bench.py's `roofline.traffic` is read from once merged into profiles/rNN/traffic.json.
"""
import json
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


# kernels whose traffic is tabulated: short label -> substring of the (shortened) kernel name
# (pp<11> / <12> / <10>: the GELU, q/k/v and residual epilogues with the LayerNorm folded in, what the forward runs)
TRAFFIC_KERNELS = {"fc1": "gemm_bf16_pp<11, 0>", "fc1_separate_ln": "gemm_bf16_pp<1, 0>", "qkv_proj": "gemm_bf16_pp<12, 0>",
                   "qkv_proj_separate_ln": "gemm_bf16_pp<3, 0>", "residual_ln_stats": "gemm_bf16_pp<10, 0>",
                   "residual_planes": "gemm_bf16_pp<14, 0>", "residual_f32_to_planes": "gemm_bf16_pp<13, 0>",
                   "residual_planes_to_f32": "gemm_bf16_pp<15, 0>",
                   "residual": "gemm_bf16_pp<2, 0>", "attention": "attention_fwd<", "layernorm": "layernorm_rows<",
                   "ln_finalize": "ln_finalize_rows"}


def main(dirs):
    workload = None
    if dirs and dirs[0] == "--workload":
        workload, dirs = dirs[1], dirs[2:]
    counters = defaultdict(dict)
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
                        counters[k][c] = sum(v) / len(v)
                    print(line)
    if workload:
        table = {}
        for label, pat in TRAFFIC_KERNELS.items():
            for k, cs in counters.items():
                if pat in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                    table[label] = {"kernel": k, "fetch_kib": cs["FETCH_SIZE"], "write_kib": cs["WRITE_SIZE"],
                                    "hbm_bytes": (2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024}
        path = os.path.join(os.path.dirname(os.path.normpath(dirs[0])), f"traffic_{workload}.json")
        with open(path, "w") as f:
            json.dump({workload: table}, f, indent=1)
        print(f"== wrote {path}")


if __name__ == "__main__":
    main(sys.argv[1:])
