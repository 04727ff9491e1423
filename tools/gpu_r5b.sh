cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shipped_models.py tests/test_gpu_cli.py -m gpu -q -x -k "native_sharded or sharded or shipped or real_weights or legacy_script or two_class or compat or whole_image" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
for sl in 2 8; do
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --force-sharded --slabs $sl > $O/bench_forced_s$sl.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench_forced_s$sl.log | tail -5 | cut -c1-1500
done
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $O/bench_plain.log 2>&1
grep "^{" $O/bench_plain.log | cut -c1-400
