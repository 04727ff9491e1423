cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
UMX_DEBUG_PLAN=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench.log 2>&1
grep "umx plan" $O/bench.log
grep -v "^W2026\|^E2026\|amdgpu.ids\|umx plan" $O/bench.log | head -22
for st in 0 10000 20000 40000; do
  UMX_STAGGER=$st timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench_st$st.log 2>&1
  echo "== stagger $st"; grep -o '"value": [0-9.]*' $O/bench_st$st.log; grep "lu0.conv \|lu0.convT\|lu1.conv \|lu4.conv " $O/bench_st$st.log
done
