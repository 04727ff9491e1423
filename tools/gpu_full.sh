# usage: bash tools/gpu_full.sh <outdir-name> -- the whole GPU test-suite + smoke, logs under gpurun_out/<name>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
