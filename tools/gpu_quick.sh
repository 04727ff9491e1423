# usage: bash tools/gpu_quick.sh <outdir-name> [pytest -k expr]  -- GPU test-suite + default bench, logs under gpurun_out/<name>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q ${2:+-k "$2"} > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench.log | tail -30
