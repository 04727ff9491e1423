# usage: bash tools/gpu_pmc2.sh <outdir-name>: stall-reason counter passes (LDS / VMEM latency, FIFO back-pressure) of one bench step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
D=/tmp/umx_prof2; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/stats -o run -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $D/p1 -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL -d $D/p2 -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES -d $D/p3 -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $O/p3.log 2>&1
N=pi2d.gather_normalise,ld0.conv,ld1.conv,ld2.conv,ld3.conv,ld4.conv,lb.conv,lu4.convT,lu4.conv,lu3.convT,lu3.conv,lu2.convT,lu2.conv,lu1.convT,lu1.conv,lu0.convT,lu0.conv,pi2d.stitch
python3 tools/summarize_rocprof.py $D/stats/run_results.db --pmc $D/p1/run_results.db $D/p2/run_results.db $D/p3/run_results.db --cycle gather_ --names $N -o $O/stall_by_layer_pmc.csv > $O/summary.log 2>&1
tail -3 $O/summary.log
