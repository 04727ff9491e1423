cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
for w in 8 4 2; do
python tools/shard_rank_timing.py --world $w > $O/new_w$w.log 2>&1; grep "^{" $O/new_w$w.log
UMX_LIB=$PWD/unmicst_amd/libumx_prev.so python tools/shard_rank_timing.py --world $w > $O/prev_w$w.log 2>&1; grep "^{" $O/prev_w$w.log
done
