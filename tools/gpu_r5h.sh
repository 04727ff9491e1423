cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "shard or band" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --force-sharded > $O/bench_forced.log 2>&1
grep "^{" $O/bench_forced.log | cut -c1-250
