#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid size): calls, average / min / max duration in us.
`--kernel-trace --stats` averages over every launch of a kernel name; bench.py launches the same kernels in
several regimes (one [T,N] block per launch AND one vec step per launch), which this keeps apart.

    python tools/trace_summary.py <..._kernel_trace.csv> [substring ...] > summary.csv"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    groups = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if pats and not any(p in name for p in pats):
                continue
            key = (short(name), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]),
                   int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["Scratch_Size"]))
            groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid_x", "grid_y", "workgroup", "lds_bytes", "vgprs", "scratch", "calls", "avg_us", "min_us", "max_us"])
    for key, d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        w.writerow(list(key) + [len(d), round(sum(d) / len(d), 3), round(min(d), 3), round(max(d), 3)])


if __name__ == "__main__":
    main()
