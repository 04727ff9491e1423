# usage: bash tools/gpu_timeline.sh <outdir-name> [bench args]: kernel + memory-copy timeline of the default (host-path) bench;
# prints the busy / idle split of the GPU's kernel activity over the last step and the largest gaps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
D=/tmp/umx_tl; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --memory-copy-trace -d $D -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $O/bench.log 2>&1
python3 - "$D/run_results.db" > $O/timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
mt = [t for t in tabs if t.startswith("memory_copies")] or [t for t in tabs if "memory_copy" in t]
print("tables:", kt[:3], mt[:3])
def cols(t): return [r[1] for r in c.execute("pragma table_info(%s)" % t)]
k = kt[0]; kc = cols(k)
name_col = "name" if "name" in kc else [x for x in kc if "name" in x][0]
rows = list(c.execute("select %s, start, end from %s order by start" % (name_col, k)))
# last step: from the last-but-one gather kernel group... take the final 40% of the trace
# the host-path phase: from the first raw -> float64 conversion to the last uint8 cast (the resident phase and the equality
# check that follow it use neither)
first = min(r[1] for r in rows if "raw_to_double" in r[0])
last = max(r[2] for r in rows if "half_to_u8" in r[0])
cut = first
sel = [r for r in rows if r[1] >= first and r[2] <= last]
busy = 0; gaps = []; cur_s, cur_e = sel[0][1], sel[0][2]; prev = sel[0]
for r in sel[1:]:
    if r[1] > cur_e:
        gaps.append((r[1] - cur_e, prev[0][:50], r[0][:50])); busy += cur_e - cur_s; cur_s, cur_e = r[1], r[2]
    else:
        cur_e = max(cur_e, r[2])
    prev = r
busy += cur_e - cur_s
span = sel[-1][2] - sel[0][1]
print("span %.3f ms, kernels busy %.3f ms (%.1f %%), idle %.3f ms in %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
for g in sorted(gaps, reverse=True)[:24]:
    print("  gap %8.1f us  after %-50s before %s" % (g[0] / 1e3, g[1], g[2]))
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r[0][:60]]; a[0] += 1; a[1] += r[2] - r[1]
print("kernel time in the window:")
for n, a in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-60s %5d %9.3f ms" % (n, a[0], a[1] / 1e6))
if mt:
    m = mt[0]; mc = cols(m)
    print("memcpy cols:", mc)
    mr = list(c.execute("select * from %s where start >= %d and end <= %d order by start" % (m, first, last)))
    si, ei = mc.index("start"), mc.index("end")
    tot = sum(r[ei] - r[si] for r in mr)
    print("memory copies in the window: %d, %.3f ms total" % (len(mr), tot / 1e6))
PY
cat $O/timeline.txt
grep '^{' $O/bench.log | cut -c1-200
