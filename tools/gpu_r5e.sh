cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s -k "baseline or finite" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^W2026\|^E2026" $O/pytest.log | tail -12 | cut -c1-600
timeout 600 python bench.py --workload train-synth256 --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_train.log 2>&1
grep "^{" $O/bench_train.log | cut -c1-300
timeout 600 python bench.py --workload train-synth256 --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_train2.log 2>&1
grep "^{" $O/bench_train2.log | cut -c1-300
