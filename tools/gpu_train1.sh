cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/train5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^FAILED" $O/pytest.log | tail -12
for i in 1 2; do timeout 600 python bench.py --workload train-synth256 --steps 30 --warmup 3 --cpu-seconds 0 > $O/bench_train$i.log 2>&1; grep "^{" $O/bench_train$i.log | cut -c1-200; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
