cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=4 -k "native_sharded" > $O/pytest_new.log 2>&1; tail -8 $O/pytest_new.log
for f in "" "--native-shard"; do timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --force-sharded $f > $O/bench_sh$f.log 2>&1; grep '^{' $O/bench_sh$f.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('sharded1 $f', j['value'], j['ms_per_step'], j['resident']['value'], j['config']['host_path_equals_resident_path'], j['config']['checksum'])"; tail -2 $O/bench_sh$f.log | cut -c1-200; done
