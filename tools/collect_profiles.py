"""Copy the summaries of gpurun_out/prof_<workload>/ (written by tools/profile.sh) into profiles/<round>/ and merge
their traffic tables into profiles/<round>/traffic.json, the file bench.py reads `roofline.traffic` from.

    python tools/collect_profiles.py r02 b16_224_bs256 l16_224_bs256 b16_384_bs128
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(rnd, workloads):
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    traffic_path = os.path.join(dst, "traffic.json")
    traffic = {}
    if os.path.exists(traffic_path):
        with open(traffic_path) as f:
            traffic = json.load(f)
    for w in workloads:
        src = os.path.join(ROOT, "gpurun_out", f"prof_{w}")
        shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"bench_{w}_kernel_stats.csv"))
        shutil.copy(os.path.join(src, "bench_line_under_rocprof.json"), os.path.join(dst, f"bench_{w}_line_under_rocprof.json"))
        shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"pmc_summary_{w}.txt"))
        with open(os.path.join(src, f"traffic_{w}.json")) as f:
            traffic.update(json.load(f))
        print("collected", w)
    with open(traffic_path, "w") as f:
        json.dump(traffic, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
