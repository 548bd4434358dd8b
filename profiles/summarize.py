#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/prof_*) into the small summaries committed under profiles/.

usage: summarize.py TAG STATS_DIR FETCH_DIR WRITE_DIR NOTE [WORKLOAD [PRECISION]]
  STATS_DIR : rocprofv3 --kernel-trace --stats --output-format csv  -- python3 bench.py ...
  FETCH_DIR : rocprofv3 --pmc FETCH_SIZE --output-format csv        -- python3 bench.py ...   (own pass)
  WRITE_DIR : rocprofv3 --pmc WRITE_SIZE --output-format csv        -- python3 bench.py ...   (own pass)
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

KIND = {"0": "fft_f_tile", "1": "fft_f_line", "5": "gain_inv", "6": "gain_line", "7": "gain_fwd", "8": "reduce",
        "9": "tail_inv", "10": "tail_line"}
# bench.py's kernel groups (KERNEL_NAMES) -> rocprof kernel kinds
GROUP = {"gain_inv": "gain_inv", "gain_line": "gain_line", "gain_fwd": "gain_fwd", "reduce": "reduce"}


def kind_of(name):
    import re
    m = re.search(r"\(bfsm::K\)(\d+)", name)
    return KIND.get(m.group(1)) if m else None


def main():
    tag, stats_dir, fetch_dir, write_dir, note = sys.argv[1:6]
    workload = sys.argv[6] if len(sys.argv) > 6 else "cfg3"
    precision = int(sys.argv[7]) if len(sys.argv) > 7 else (32 if workload == "cfg5" else 64)
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))
    out = {}
    for r in csv.DictReader(open(stats)):
        k = kind_of(r["Name"])
        if k:
            out.setdefault(k, {})["avg_launch_us"] = float(r["AverageNs"]) / 1e3
            out[k]["calls"] = int(r["Calls"])
    for cname, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = kind_of(r["Kernel_Name"])
            if k:
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            out.setdefault(k, {})[cname + "_KiB_per_launch"] = v / n
    for k, d in out.items():
        d["hbm_bytes_per_launch"] = (2 * d.get("FETCH_SIZE_KiB_per_launch", 0) + d.get("WRITE_SIZE_KiB_per_launch", 0)) * 1024
        if d.get("avg_launch_us"):
            d["hbm_GBps"] = d["hbm_bytes_per_launch"] / (d["avg_launch_us"] * 1e-6) / 1e9
    out["_note"] = note
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    latest = {g: {"hbm_bytes_per_launch": out[k]["hbm_bytes_per_launch"]} for g, k in GROUP.items() if k in out}
    latest["_source"] = f"{tag}_pmc_traffic.json"
    # bench.py reports this traffic only for the workload and the kernel build it was collected on
    latest["_workload"], latest["_precision"] = workload, precision
    sys.path.insert(0, os.path.dirname(here))
    import bench
    latest["_kernel_src"] = out["_kernel_src"] = bench.kernel_source_digest()
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    if workload == "cfg3":      # the default bench command is the one whose roofline object quotes the traffic
        json.dump(latest, open(os.path.join(here, "traffic_latest.json"), "w"), indent=1, sort_keys=True)
    for k in ("gain_inv", "gain_line", "gain_fwd", "reduce"):
        if k in out:
            print(k, {a: round(b, 1) for a, b in out[k].items()})


if __name__ == "__main__":
    main()
