cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -q > $O/pytest_train.log 2>&1; tail -1 $O/pytest_train.log
timeout 900 python bench.py --workload train-synth256 --steps 50 --warmup 5 --cpu-seconds 30 > $O/bench_train.log 2>&1
bash tools/gpu_train_prof.sh final/train_prof 8 > $O/train_prof.txt 2>&1
bash tools/gpu_train_pmc.sh final/train_pmc > $O/train_pmc.txt 2>&1
grep "^{" $O/bench_train.log | cut -c1-250
