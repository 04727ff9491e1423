# usage: bash tools/gpu_r3_final.sh -- the round's final evidence: GPU test-suite, default bench line, per-layer breakdowns, rocprofv3
# kernel stats of the default command + separate PMC passes, the other workloads' lines.  Outputs under gpurun_out/final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --resident-only --breakdown > $O/bench_f16x3_breakdown.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --scaling strong > $O/bench_strong_n1.log 2>&1
for w in solo-16384 duo-4096 solo-1024 legacy-1024; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 30 --breakdown > $O/bench_$w.log 2>&1
done
bash tools/gpu_pmc.sh final/pmc_synth256 > $O/pmc_synth256.log 2>&1
tail -3 $O/pytest_gpu.log
for f in $O/bench_*.log; do echo == $f; grep "^{" $f | cut -c1-220; done
