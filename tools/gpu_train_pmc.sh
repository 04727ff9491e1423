# usage: bash tools/gpu_train_pmc.sh <outdir-name> -- PMC passes of the training bench (batch 8, 2 steps), summarised per (kernel, grid):
# matrix-pipe busy cycles, wait / active cycles, HBM bytes (separate passes, as MI355X_MICROARCH.md prescribes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
D=/tmp/umx_tprof; rm -rf $D; mkdir -p $D
A="--workload train-synth256 --steps 2 --warmup 1 --cpu-seconds 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $D/pmc_sq -o run -- python3 bench.py $A > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/pmc_fetch -o run -- python3 bench.py $A > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/pmc_write -o run -- python3 bench.py $A > $O/pmc_write.log 2>&1
python3 tools/summarize_rocprof.py $D/pmc_sq/run_results.db --pmc $D/pmc_sq/run_results.db $D/pmc_fetch/run_results.db $D/pmc_write/run_results.db -o $O/train_b8_by_kernel_grid_pmc.csv
wc -l $O/train_b8_by_kernel_grid_pmc.csv
