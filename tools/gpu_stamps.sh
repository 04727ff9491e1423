cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for l in lu0.conv lu0.convT lu1.conv ld0.conv lu4.conv lu2.conv; do
  UMX_DEBUG_STAMPS=$l timeout 300 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep "umx stamps" | tail -2
done
