cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for mb in 0 80 160 320 640; do
  UMX_CHAIN_MB=$mb timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench_chain$mb.log 2>&1
  echo "== chain $mb"; grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench_chain$mb.log | grep -o '"value": [0-9.]*'
done
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench_chain160.log | head -24
