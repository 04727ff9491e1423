# usage: bash tools/gpu_train_prof.sh <outdir-name> [train-batch] -- rocprofv3 kernel trace of the training bench, summarised per (kernel, grid)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
B=${2:-8}
timeout 600 python bench.py --workload train-synth256 --train-batch $B --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_train.log 2>&1
grep '^{' $O/bench_train.log | tail -1
rm -rf /tmp/prof_t; mkdir -p /tmp/prof_t
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python3 bench.py --workload train-synth256 --train-batch $B --steps 3 --warmup 1 --cpu-seconds 0 > $O/rocprof_train.log 2>&1
DB=$(find /tmp/prof_t -name '*results.db' | head -1)
python tools/summarize_rocprof.py $DB -o $O/train_b${B}_by_kernel_grid.csv > $O/summary.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_b${B}_by_kernel_grid.csv")))
tot=sum(float(r["total_ms"]) for r in rows)
agg={}
for r in rows:
    k=r["kernel"].split("<")[0]
    agg[k]=agg.get(k,0)+float(r["total_ms"])
print("total kernel ms", round(tot,2))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]): print("%-40s %9.3f ms %5.1f%%"%(k,v,100*v/tot))
print()
for r in sorted(rows, key=lambda r:-float(r["total_ms"]))[:40]:
    print("%-44s wg %7s y %3s z %2s lds %6s vgpr %4s calls %4s avg_us %9s total_ms %8s"%(r["kernel"][:44],r["workgroups_x"],r["grid_y"],r["grid_z"],r["lds_bytes"],r["vgprs"],r["calls"],r["avg_us"],r["total_ms"]))
PY
