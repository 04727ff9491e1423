cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for cap in 81920 54272 40960; do
  UMX_LDS_CAP_NARROW=$cap timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench_cap$cap.log 2>&1
  echo "== cap $cap"; grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench_cap$cap.log | grep -o '"value": [0-9.]*'
  grep "NT=3" $O/bench_cap$cap.log
done
