cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench.log | head -22
grep -o '"value": [0-9.]*' $O/bench.log
for l in lu0.conv lu0.convT lu1.conv ld0.conv lu1.convT; do
  UMX_DEBUG_STAMPS=$l timeout 300 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep "umx stamps" | head -1
done
