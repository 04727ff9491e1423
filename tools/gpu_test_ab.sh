# usage: bash tools/gpu_test_ab.sh <outdir-name> <variant-file> [pytest -k expr]: parity tests first, then the A/B bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x ${3:+-k "$3"} > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
bash tools/gpu_ab.sh $1 $2
