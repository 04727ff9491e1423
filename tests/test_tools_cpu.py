"""The evidence tools that turn rocprofv3 databases into the tables under profiles/ (tools/summarize_rocprof.py,
tools/timeline_rocprof.py), run on a small hand-made rocpd-shaped database: grouping by (kernel, grid), the demangling of names
rocprofv3 leaves mangled, and the per-stream time line of the last step."""
import csv
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path):
    c = sqlite3.connect(path)
    c.execute("create table kernels (name text, start integer, end integer, duration integer, grid_x integer, grid_y integer, "
              "grid_z integer, workgroup_x integer, lds_size integer, vgpr_count integer, accum_vgpr_count integer, stream_id integer)")
    rows, t = [], 1000
    for step in range(3):
        for name, dur, gx, stream in [("void umx::conv_f16x3<9, 4, 1>(umx::HConvParams)", 700, 256 * 32, 1),
                                      ("_ZN3umx16split_dyn_kernelEPKfmiiPKjPfPDF16_S5_PiPj", 40, 256 * 4096, 1),
                                      ("void umx::wgrad_f16x3<true>(umx::WgradParams)", 250, 256 * 171, 2),
                                      ("void at::native::elementwise_kernel<128>(int)", 5, 256, 1),
                                      ("void umx::softmax_loss_kernel(float const*)", 11, 256 * 1024, 1)]:
            rows.append((name, t, t + dur, dur, gx, 1, 1, 256, 80000, 128, 0, stream))
            t += dur + (7 if stream == 1 else 0)
    c.executemany("insert into kernels values (?,?,?,?,?,?,?,?,?,?,?,?)", rows)
    c.commit()
    c.close()


def test_summary_groups_by_kernel_and_grid_and_demangles(tmp_path):
    db = str(tmp_path / "run_results.db")
    _db(db)
    out = str(tmp_path / "by_kernel.csv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), db, "-o", out], check=True)
    rows = list(csv.DictReader(open(out)))
    names = {r["kernel"] for r in rows}
    assert "umx::split_dyn_kernel" in names                       # (left mangled by rocprofv3: _Float16 parameters)
    assert not any("elementwise" in n for n in names)             # only umx:: kernels unless --all
    conv = next(r for r in rows if r["kernel"].startswith("umx::conv_f16x3"))
    assert conv["calls"] == "3" and conv["workgroups_x"] == "32" and float(conv["avg_us"]) == 0.7


def test_time_line_of_the_last_step_with_streams(tmp_path):
    db = str(tmp_path / "run_results.db")
    _db(db)
    out = str(tmp_path / "tl.txt")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline_rocprof.py"), db, "-o", out], check=True)
    text = open(out).read().splitlines()
    body = [l for l in text if l and not l.startswith("#") and not l.lstrip().startswith("start_us")]
    assert len(body) == 5                                          # the dispatches between the last two loss kernels
    assert body[-1].split()[4].startswith("umx::softmax_loss_kernel")
    streams = {l.split()[3] for l in body}
    assert streams == {"1", "2"}
    busy = [l for l in text if l.startswith("# stream")]
    assert len(busy) == 2 and busy[0].startswith("# stream 1")     # sorted by busy time
