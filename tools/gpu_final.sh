# Round-end measurement set: GPU test-suite, the default bench line (host path + resident + CPU baselines), per-layer
# breakdowns of both precisions, the other BASELINE configs, rocprofv3 kernel stats + PMC passes of the default configuration,
# the training step.  Summaries land in gpurun_out/<name>/ (copy the ones to keep to profiles/rNN/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -14 $O/pytest_gpu.log
( time python bench.py ) > $O/bench_default.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench_f16x3_breakdown.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --breakdown --precision f32 > $O/bench_f32_breakdown.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 10 --workload solo-16384 --batch 1024 > $O/bench_solo16384.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload solo-1024 --batch 484 > $O/bench_solo1024.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 10 --workload duo-4096 > $O/bench_duo4096.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 10 --workload legacy-1024 --batch 121 > $O/bench_legacy1024.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --force-sharded > $O/bench_sharded_world1.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --force-sharded --native-shard > $O/bench_sharded_world1_native.log 2>&1
for f in bench_default bench_solo16384 bench_solo1024 bench_duo4096 bench_legacy1024 bench_sharded_world1 bench_sharded_world1_native; do grep -v "^W2026\|^E2026\|amdgpu.ids" $O/$f.log | grep '^{' | tail -1 | cut -c1-330; done
bash tools/gpu_pmc.sh $1/pmc --resident-only
timeout 600 python bench.py --workload train-synth256 --steps 10 --warmup 2 > $O/bench_train.log 2>&1
grep '^{' $O/bench_train.log | tail -1 | cut -c1-400
