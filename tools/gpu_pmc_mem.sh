# usage: bash tools/gpu_pmc_mem.sh <outdir-name> [bench args]: texture-address / L1 / L2 counters of one bench step, per layer.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
D=/tmp/umx_prof_m; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/stats -o run -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE -d $D/pmc_a -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -d $D/pmc_b -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum -d $D/pmc_c -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_c.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_BUSY_sum TCC_REQ_sum TCC_TAG_STALL_sum -d $D/pmc_d -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_d.log 2>&1
N=pi2d.gather_normalise,ld0.conv,ld1.conv,ld2.conv,ld3.conv,ld4.conv,lb.conv,lu4.convT,lu4.conv,lu3.convT,lu3.conv,lu2.convT,lu2.conv,lu1.convT,lu1.conv,lu0.convT,lu0.conv,pi2d.stitch
python3 tools/summarize_rocprof.py $D/stats/run_results.db --pmc $D/pmc_a/run_results.db $D/pmc_b/run_results.db $D/pmc_c/run_results.db $D/pmc_d/run_results.db --cycle gather_ --names $N -o $O/by_layer_mem.csv
for f in $O/*.log; do echo == $f; grep -v "^W2026\|^E2026\|amdgpu.ids" $f | tail -2 | cut -c1-200; done
