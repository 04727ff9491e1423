# usage: bash tools/gpu_quicktest.sh <outdir-name> "<env>"   -- a quick f16x3 parity subset (small graphs, shipped hp)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
env $2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=3 -k "(forward_tiles_matches_oracle or shipped_hyper) and f16x3" > $O/quick_$(echo $2 | tr -c 'A-Za-z0-9' _).log 2>&1
tail -12 $O/quick_*.log | grep -E "passed|failed|error|Error|assert" | head -12
