# usage: bash tools/gpu_stamps.sh <outdir-name> "<env>" layer...   -- in-kernel s_memtime segments of the named layers
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
E="$1"; shift
for L in "$@"; do
  env $E UMX_DEBUG_STAMPS=$L timeout 300 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep "umx stamps" | tail -1 | sed "s/^/[$E] /" >> $O/stamps.log
done
cat $O/stamps.log
