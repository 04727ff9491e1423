# usage: bash tools/gpu_r4_mid.sh -- mid-round check: GPU suite, the default line (16384^2 slide), a world-of-one run of the N > 1 path
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4mid; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 600 python bench.py --force-sharded --steps 3 --warmup 1 --cpu-seconds 0 --scaling weak > $O/bench_forced_sharded.log 2>&1
for f in $O/bench_*.log; do echo == $f; grep "^{" $f | cut -c1-400; done
