#!/usr/bin/env python3
"""One training step as a time line, from a rocprofv3 rocpd database (`*_results.db`): every dispatch between the last two
launches of a marker kernel (default: softmax_loss_kernel closes a forward pass), in start order, with its stream / queue, start
offset, duration, and the gap since the previous dispatch ended ON THE SAME STREAM.  The per-stream busy time and the idle gaps on
the main stream are what the per-kernel totals of summarize_rocprof.py cannot show.
usage: timeline_rocprof.py results.db [--marker NAME] [-o out.txt]"""
import argparse
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_rocprof import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--marker", default="softmax_loss_kernel")
    ap.add_argument("-o", default=None)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = list(c.execute("select name, start, end, grid_x / workgroup_x, grid_y, grid_z, %s from kernels order by start" % sid))
    marks = [i for i, r in enumerate(rows) if a.marker in r[0]]
    if len(marks) < 2:
        sys.exit("fewer than two %s launches in the trace" % a.marker)
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    out = open(a.o, "w") if a.o else sys.stdout
    last_end, busy = {}, {}
    print("# columns of table kernels: %s" % ", ".join(cols), file=out)
    print("# one step = %d dispatches, %.3f ms from the first start to the last end" % (
        len(step), (max(r[2] for r in step) - t0) / 1e6), file=out)
    print("%9s %8s %8s %6s  %-46s %s" % ("start_us", "dur_us", "gap_us", "stream", "kernel", "grid"), file=out)
    for name, s, e, gx, gy, gz, st in step:
        gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
        last_end[st] = max(e, last_end.get(st, 0))
        busy[st] = busy.get(st, 0) + (e - s)
        print("%9.1f %8.1f %8.1f %6s  %-46s %dx%dx%d" % ((s - t0) / 1e3, (e - s) / 1e3, gap, st, short(name)[:46], gx, gy, gz), file=out)
    for st, b in sorted(busy.items(), key=lambda kv: -kv[1]):
        print("# stream %s busy %.3f ms" % (st, b / 1e6), file=out)


if __name__ == "__main__":
    main()
