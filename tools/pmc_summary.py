#!/usr/bin/env python3
"""Compact summary of a tools/gpu_profile.sh output directory (runs on the GPU box, keeps only our kernels).

    python tools/pmc_summary.py gpurun_out/prof_TAG  ->  gpurun_out/prof_TAG_summary.json + _kernel_stats.csv
"""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def ours(name):
    return "k_threshold" in name or "::k_" in name or "k_png" in name


def short(name):
    for k in ("k_threshold<2, false, 7>", "k_threshold<2, false, 6>", "k_threshold<2, false, 2>", "k_threshold<2, false, 3>", "k_threshold<3, true, 2>", "k_threshold<3, true, 3>",
              "k_threshold<2, false>", "k_threshold<3, true>", "k_rs<4>", "k_rs<2>", "k_symbols", "k_flood_wave", "k_flood3", "k_frame_mid", "k_colors",
              "k_frame_end", "k_carry_out", "k_count_flagged", "k_plane_bytes", "k_png_inflate4<2048>", "k_png_inflate<8192, true>", "k_png_inflate<8192, false>",
              "k_png_inflate<32768, true>", "k_png_inflate<32768, false>", "k_png_unfilter"):
        if k in name:
            return k
    return name[:60]


def main():
    d = sys.argv[1].rstrip("/")
    out = {"kernel_stats": [], "pmc": {}}
    for path in glob.glob(os.path.join(d, "trace", "*kernel_stats.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                if ours(row.get("Name", "")):
                    out["kernel_stats"].append({"kernel": short(row["Name"]), "calls": int(row["Calls"]),
                                                "total_ns": float(row["TotalDurationNs"]), "avg_ns": float(row["AverageNs"]),
                                                "min_ns": float(row.get("MinNs", 0)), "max_ns": float(row.get("MaxNs", 0)),
                                                "pct": float(row.get("Percentage", 0))})
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for path in glob.glob(os.path.join(d, "pmc*", "*counter_collection.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                if not ours(name):
                    continue
                k = short(name)
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta[k] = {"vgpr": row.get("VGPR_Count"), "sgpr": row.get("SGPR_Count"), "lds": row.get("LDS_Block_Size"),
                           "scratch": row.get("Scratch_Size"), "grid": row.get("Grid_Size"), "wg": row.get("Workgroup_Size")}
    for k, cs in acc.items():
        out["pmc"][k] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
        out["pmc"][k]["_dispatches_averaged"] = len(next(iter(cs.values())))
        out["pmc"][k].update({"_" + a: b for a, b in meta[k].items()})
    with open(d + "_summary.json", "w") as f:
        json.dump(out, f, indent=1)
    with open(d + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_of_all_gpu_kernel_time"])
        for r in sorted(out["kernel_stats"], key=lambda r: -r["total_ns"]):
            w.writerow([r["kernel"], r["calls"], r["total_ns"], r["avg_ns"], r["min_ns"], r["max_ns"], r["pct"]])
    print(json.dumps(out, indent=1)[:6000])


if __name__ == "__main__":
    main()
