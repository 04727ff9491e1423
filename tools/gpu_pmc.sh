# usage: bash tools/gpu_pmc.sh <outdir-name> [bench args]: rocprofv3 kernel stats + PMC passes of one bench step.
# The rocpd databases stay in /tmp on the box; only the per-layer CSV summaries land in gpurun_out/<name>/.
# The --stats pass profiles the DEFAULT command (host path + synchronous call + resident slide); the counter passes run the resident
# slide only, so that every UNet pass has the same launch sequence and the per-layer table lines up (same kernels, same shapes).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
D=/tmp/umx_prof; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/stats -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-legs "$@" > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $D/pmc_sq -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --resident-only "$@" > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $D/pmc_tcc -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --resident-only "$@" > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --resident-only "$@" > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/pmc_write -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --resident-only "$@" > $O/pmc_write.log 2>&1
# one UNet pass of the split-precision plan = gather (+split), 16 convs (the softmax head is fused into the last one)
# (UMX_PMC_NAMES overrides the labels: the solo graph has four down-sampling levels, not five)
N=${UMX_PMC_NAMES:-pi2d.gather_normalise,ld0.conv,ld1.conv,ld2.conv,ld3.conv,ld4.conv,lb.conv,lu4.convT,lu4.conv,lu3.convT,lu3.conv,lu2.convT,lu2.conv,lu1.convT,lu1.conv,lu0.convT,lu0.conv,pi2d.stitch}
python3 tools/summarize_rocprof.py $D/pmc_sq/run_results.db --pmc $D/pmc_sq/run_results.db $D/pmc_tcc/run_results.db $D/pmc_fetch/run_results.db $D/pmc_write/run_results.db --cycle gather_ --names $N -o $O/by_layer_pmc.csv
python3 tools/summarize_rocprof.py $D/stats/run_results.db -o $O/by_kernel_grid.csv
python3 - <<PY
import sqlite3, csv
c = sqlite3.connect("$D/stats/run_results.db")
w = csv.writer(open("$O/kernel_stats.csv", "w", newline=""))
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for r in c.execute("select * from top_kernels"):
    w.writerow(r)
PY
for f in $O/*.log; do echo == $f; grep -v "^W2026\|^E2026\|amdgpu.ids" $f | tail -2 | cut -c1-300; done
