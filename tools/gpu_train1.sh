cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/train3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^FAILED" $O/pytest.log | tail -12
timeout 600 python bench.py --workload train-synth256 --steps 30 --warmup 3 --cpu-seconds 0 > $O/bench_train.log 2>&1; grep "^{" $O/bench_train.log | cut -c1-260
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --scaling weak --resident-only --breakdown > $O/bench_inf.log 2>&1; grep "^{" $O/bench_inf.log | cut -c1-200
