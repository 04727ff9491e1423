# Round-end measurement set: GPU test-suite, the default bench line (with CPU baseline), both precisions, rocprofv3
# kernel stats + PMC passes of the default configuration.  Summaries land in gpurun_out/<name>/ (copy to profiles/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
( time python bench.py ) > $O/bench_default.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench_f16x3_breakdown.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --breakdown --precision f32 > $O/bench_f32_breakdown.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload solo-1024 --batch 484 > $O/bench_solo1024.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload duo-4096 > $O/bench_duo4096.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload legacy-1024 --batch 121 > $O/bench_legacy1024.log 2>&1
for f in bench_default bench_solo1024 bench_duo4096 bench_legacy1024; do grep -v "^W2026\|^E2026\|amdgpu.ids" $O/$f.log | tail -5 | cut -c1-400; done
bash tools/gpu_pmc.sh $1/pmc
# training step (BASELINE configs[4]): bench line with CPU baseline, kernel trace summary, parity report
timeout 600 python bench.py --workload train-synth256 --steps 10 --warmup 2 > $O/bench_train.log 2>&1
grep '^{' $O/bench_train.log | tail -1 | cut -c1-600
bash tools/gpu_train_prof.sh $1/train 8 > $O/train_b8_summary.txt 2>&1
timeout 600 python tests/train_parity_report.py 2>&1 | grep -v Warn > $O/train_parity_report.log
