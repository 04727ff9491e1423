# usage: bash tools/gpu_tests_and_breakdown.sh <outdir-name> [pytest -k expr]  -- GPU test-suite, then the resident per-layer breakdown of the default workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q -x ${2:+-k "$2"}) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/pytest_gpu.log | tail -40
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --resident-only --scaling weak --breakdown > $O/bench_breakdown.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench_breakdown.log | tail -45
