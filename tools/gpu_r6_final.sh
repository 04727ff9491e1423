# usage: bash tools/gpu_r6_final.sh -- the round's final evidence on one box: GPU test-suite, smoke, the default bench line (16384^2 slide, with its
# exact-fp32, training and BASELINE-config legs), the same slide through the N > 1 schedule in a world of one (--force-sharded), per-layer breakdowns
# of the default precision (fp6 cross terms on the layers that gain) next to the pure 3-product one and to the form on every eligible layer, the training line, rocprofv3 kernel stats of the default command and the separate
# PMC passes (on the 2048-row band: the same launch sequence four times over).  Outputs under gpurun_out/final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 600 python bench.py --force-sharded --cpu-seconds 0 > $O/bench_force_sharded.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --resident-only --scaling weak --breakdown --precision f16x3 > $O/bench_f16x3_breakdown.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --resident-only --scaling weak --breakdown > $O/bench_f16f6_breakdown.log 2>&1
UMX_F6_ALL=1 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --resident-only --scaling weak --breakdown > $O/bench_f16f6_all_layers_breakdown.log 2>&1
timeout 900 python bench.py --workload train-synth256 --steps 50 --warmup 5 --cpu-seconds 30 > $O/bench_train.log 2>&1
timeout 600 python tests/f16f6_parity_report.py > $O/f16f6_parity.log 2>&1
bash tools/gpu_pmc.sh final/pmc_synth256 --scaling weak > $O/pmc_synth256.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; tail -1 $O/smoke.log
for f in $O/bench_*.log; do echo == $f; grep "^{" $f | cut -c1-200; done
