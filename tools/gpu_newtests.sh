cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=8 -k "range_overflow or clean_checkout or duo_script or solo_full_16384 or sharded" > $O/pytest_new.log 2>&1; tail -16 $O/pytest_new.log
