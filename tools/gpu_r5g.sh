cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_switches.py tests/test_gpu_cli.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for b in 256 384 512; do
timeout 600 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --resident-only --no-legs --batch $b --breakdown > $O/bench_b$b.log 2>&1
grep "^{" $O/bench_b$b.log | cut -c1-200; grep "pi2d.stitch\|pi2d.gather" $O/bench_b$b.log
done
