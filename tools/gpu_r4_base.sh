# usage: bash tools/gpu_r4_base.sh -- round-4 starting point on one box: GPU suite, default line, strong line, train line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4base; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --breakdown > $O/bench_default.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --scaling strong > $O/bench_strong.log 2>&1
timeout 600 python bench.py --workload train-synth256 --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_train.log 2>&1
tail -3 $O/pytest_gpu.log
for f in $O/bench_*.log; do echo == $f; grep "^{" $f | cut -c1-300; done
