cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_cli.py -m gpu -q -x --durations=4 > $O/pytest_new.log 2>&1; tail -12 $O/pytest_new.log
