# usage: bash tools/gpu_train_timeline.sh <outdir-name> [train-batch] -- one training step as a time line with stream ids
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
B=${2:-8}
rm -rf /tmp/prof_tl; mkdir -p /tmp/prof_tl
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_tl -o t -- python3 bench.py --workload train-synth256 --train-batch $B --steps 3 --warmup 1 --cpu-seconds 0 > $O/rocprof_tl.log 2>&1
DB=$(find /tmp/prof_tl -name '*results.db' | head -1)
python tools/timeline_rocprof.py $DB -o $O/train_b${B}_timeline.txt
tail -5 $O/train_b${B}_timeline.txt
