# usage: bash tools/gpu_pmc_inst.sh <outdir-name> [bench args]: instruction-mix PMC passes of one bench step, per layer.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
D=/tmp/umx_prof_i; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/stats -o run -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $D/pmc_a -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $D/pmc_b -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU -d $D/pmc_c -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/pmc_c.log 2>&1
N=pi2d.gather_normalise,ld0.conv,ld1.conv,ld2.conv,ld3.conv,ld4.conv,lb.conv,lu4.convT,lu4.conv,lu3.convT,lu3.conv,lu2.convT,lu2.conv,lu1.convT,lu1.conv,lu0.convT,lu0.conv,pi2d.stitch
python3 tools/summarize_rocprof.py $D/stats/run_results.db --pmc $D/pmc_a/run_results.db $D/pmc_b/run_results.db $D/pmc_c/run_results.db --cycle gather_ --names $N -o $O/by_layer_inst.csv
for f in $O/*.log; do echo == $f; grep -v "^W2026\|^E2026\|amdgpu.ids" $f | tail -2 | cut -c1-300; done
