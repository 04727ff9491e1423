#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into a per-(kernel, grid) table.

`rocprofv3 --stats` groups by kernel NAME, and one conv_mfma_f32<NT,HPIX> instantiation serves several UNet layers,
so the stock *_kernel_stats.csv mixes layers.  Grid size identifies the layer (bench.py --breakdown prints the same
layers from the in-library HIP events), so this groups by (name, grid) and prints calls / avg / total.
usage: summarize_rocprof.py <kernel_trace.csv> [out.csv]
"""
import csv
import sys
from collections import OrderedDict


def main():
    rows = OrderedDict()
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "umx::" not in name:
                continue
            name = name.replace("void ", "").split("(")[0]
            key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
                   int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e = rows.setdefault(key, [0, 0, 1 << 62, 0])
            e[0] += 1
            e[1] += d
            e[2] = min(e[2], d)
            e[3] = max(e[3], d)
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "workgroups_x", "grid_y", "grid_z", "lds_bytes", "vgprs", "calls", "avg_us", "min_us", "max_us",
                "total_ms"])
    for k, e in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        w.writerow(list(k) + [e[0], round(e[1] / e[0] / 1e3, 2), round(e[2] / 1e3, 2), round(e[3] / 1e3, 2),
                              round(e[1] / 1e6, 3)])


if __name__ == "__main__":
    main()
