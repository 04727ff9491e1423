# round-2 first GPU call: baseline bench + per-layer breakdown + stamps, RCCL duplicate-GPU probe
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --breakdown > $O/bench.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/bench.log | tail -30
for L in lu0.conv lu1.conv lu4.conv lu0.convT; do
  UMX_DEBUG_STAMPS=$L timeout 300 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep "umx stamps" | tail -1 >> $O/stamps.log
done
cat $O/stamps.log
for v in "" "NCCL_IGNORE_DUPLICATE_GPU=1" ; do
  echo "== env: $v" >> $O/rccl_dup.log
  env $v timeout 180 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/probes/rccl_dup.py >> $O/rccl_dup.log 2>&1
  echo "rc=$?" >> $O/rccl_dup.log
done
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/rccl_dup.log | tail -30
