cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for s in 99; do UMX_HOST_SLABS=$s timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $O/bench_slabs$s.log 2>&1; grep '^{' $O/bench_slabs$s.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('slabs $s', j['value'], j['ms_per_step'], j['resident']['value'], j['resident']['ms_per_step'], j['config']['host_path_equals_resident_path'])"; tail -3 $O/bench_slabs$s.log | cut -c1-300; done
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py -m gpu -q -x -k "raw or cli or edge" --durations=4 > $O/pytest_host.log 2>&1; tail -4 $O/pytest_host.log
