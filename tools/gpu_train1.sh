cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/train4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^FAILED" $O/pytest.log | tail -12
for i in 1 2; do timeout 600 python bench.py --workload train-synth256 --steps 30 --warmup 3 --cpu-seconds 0 > $O/bench_train$i.log 2>&1; grep "^{" $O/bench_train$i.log | cut -c1-200; done
UMX_TRAIN_NO_OVERLAP=1 timeout 600 python bench.py --workload train-synth256 --steps 30 --warmup 3 --cpu-seconds 0 > $O/bench_train_noov.log 2>&1; grep "^{" $O/bench_train_noov.log | cut -c1-200
