#!/usr/bin/env python3
"""Condense rocprofv3 output into a per-(kernel, grid) table.

`rocprofv3 --stats` groups by kernel NAME, and one conv kernel instantiation serves several UNet layers, so the stock
*_kernel_stats.csv mixes layers.  Grid size identifies the layer (bench.py --breakdown prints the same layers from
the in-library HIP events), so this groups by (name, grid) and prints calls / avg / total.

Inputs: a --kernel-trace CSV (``*_kernel_trace.csv``) or a rocpd database (``*_results.db``, the default output
format of rocprofv3 on ROCm 7.2).  With ``--pmc DB [DB ...]`` the per-dispatch counter values of separate counter
passes (FETCH_SIZE, WRITE_SIZE: KiB per dispatch) are averaged per (kernel, grid) and appended as columns; the
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of wide coalesced
reads) is applied in the ``hbm_read_MB_corrected`` column.

With ``--cycle SUBSTR`` dispatches are additionally keyed by their position since the last kernel whose name
contains SUBSTR (e.g. ``gather_normalise``: one UNet pass = gather, split, conv x16, head), which separates layers that
share a (kernel, grid) pair; ``--names a,b,c`` labels the positions.

usage: summarize_rocprof.py <kernel_trace.csv | results.db> [--pmc results.db ...] [--cycle SUBSTR] [-o out.csv]
"""
import argparse
import csv
import sqlite3
import sys
from collections import OrderedDict


def short(name):
    if name.startswith("_ZN3umx"):   # (rocprofv3 leaves names with _Float16 parameters mangled: _ZN3umx16split_dyn_kernelE...)
        import re
        m = re.match(r"_ZN3umx(\d+)", name)
        if m:
            n = int(m.group(1))
            return "umx::" + name[len(m.group(0)):len(m.group(0)) + n]
    return name.replace("void ", "").split("(")[0]


def ours(name):
    return "umx::" in name or name.startswith("_ZN3umx")


def rows_from_csv(path):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            yield (r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]),
                   int(r["Grid_Size_Z"]), int(r["LDS_Block_Size"]),
                   int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0),
                   int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def rows_from_db(path):
    c = sqlite3.connect(path)
    q = ("select name, grid_x / workgroup_x, grid_y, grid_z, lds_size, vgpr_count + accum_vgpr_count, duration "
         "from kernels order by start")
    for r in c.execute(q):
        yield r


class Cycle:
    """position of a dispatch since the last kernel whose name contains `marker` (-1 before the first one)"""

    def __init__(self, marker):
        self.marker, self.pos = marker, -1

    def step(self, name):
        if not self.marker:
            return 0
        if self.marker in name:
            self.pos = 0
        elif self.pos >= 0:
            self.pos += 1
        return self.pos


def pmc_from_db(path, marker):
    """-> {counter: {(pos, kernel, wg_x, grid_y, grid_z): [sum, n]}}"""
    c = sqlite3.connect(path)
    out = {}
    q = ("select counter_name, kernel_name, grid_size_x / workgroup_size_x, grid_size_y, grid_size_z, value, dispatch_id "
         "from counters_collection order by dispatch_id, counter_name")
    cyc, last = Cycle(marker), None
    pos = 0
    for cn, kn, gx, gy, gz, v, did in c.execute(q):
        if did != last:
            pos = cyc.step(kn)
            last = did
        if not ours(kn):
            continue
        e = out.setdefault(cn, {}).setdefault((pos, short(kn), gx, gy, gz), [0.0, 0])
        e[0] += v
        e[1] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("-o", "--out")
    ap.add_argument("--all", action="store_true", help="keep kernels outside the umx:: namespace")
    ap.add_argument("--cycle", default="", help="kernel-name substring that starts a cycle of dispatches")
    ap.add_argument("--names", default="", help="comma-separated labels of the cycle positions")
    a = ap.parse_args()
    src = rows_from_db(a.trace) if a.trace.endswith(".db") else rows_from_csv(a.trace)
    rows = OrderedDict()
    cyc = Cycle(a.cycle)
    for name, gx, gy, gz, lds, vg, d in src:
        pos = cyc.step(name)
        if not ours(name) and not a.all:
            continue
        e = rows.setdefault((pos, short(name), gx, gy, gz, lds, vg), [0, 0, 1 << 62, 0])
        e[0] += 1
        e[1] += d
        e[2] = min(e[2], d)
        e[3] = max(e[3], d)
    pmc = {}
    for p in a.pmc:
        pmc.update(pmc_from_db(p, a.cycle))
    names = a.names.split(",") if a.names else []
    counters = sorted(pmc)
    out = open(a.out, "w", newline="") if a.out else sys.stdout
    w = csv.writer(out)
    head = ["pos", "kernel", "workgroups_x", "grid_y", "grid_z", "lds_bytes", "vgprs", "calls", "avg_us", "min_us", "max_us",
            "total_ms"] + ["%s_avg%s" % (c, "_KiB" if c.endswith("_SIZE") else "") for c in counters]
    if "FETCH_SIZE" in pmc:
        head.append("hbm_read_MB_corrected")
    if "WRITE_SIZE" in pmc:
        head.append("hbm_write_MB")
    w.writerow(head)
    for k, e in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        line = [names[k[0]] if 0 <= k[0] < len(names) else k[0]] + list(k[1:]) + [e[0], round(e[1] / e[0] / 1e3, 2), round(e[2] / 1e3, 2), round(e[3] / 1e3, 2),
                          round(e[1] / 1e6, 3)]
        vals = {}
        for c in counters:
            s = pmc[c].get(k[:5])
            vals[c] = s[0] / s[1] if s else None
            line.append(round(vals[c], 1) if s else "")
        if "FETCH_SIZE" in pmc:
            line.append(round(2 * vals["FETCH_SIZE"] * 1024 / 1e6, 2) if vals["FETCH_SIZE"] is not None else "")
        if "WRITE_SIZE" in pmc:
            line.append(round(vals["WRITE_SIZE"] * 1024 / 1e6, 2) if vals["WRITE_SIZE"] is not None else "")
        w.writerow(line)


if __name__ == "__main__":
    main()
