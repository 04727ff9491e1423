set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/f16a; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "test_forward_tiles_matches_oracle and f16x3 and v2_solo_like" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -30 $O/t1.log
if grep -q "rc=0" $O/t1.log; then
  timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
  tail -40 $O/pytest_gpu.log
  timeout 600 python bench.py --steps 3 --warmup 1 --precision f16x3 --cpu-seconds 0 --breakdown > $O/bench_f16x3.log 2>&1
  timeout 600 python bench.py --steps 3 --warmup 1 --precision f32 --cpu-seconds 0 --breakdown > $O/bench_f32.log 2>&1
  tail -30 $O/bench_f16x3.log
fi
