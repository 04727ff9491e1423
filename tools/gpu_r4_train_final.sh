# usage: bash tools/gpu_r4_train_final.sh -- the training evidence only (GPU train tests, bench lines at batch 8 / 16 / 32, kernel trace, per-stream
# time line, PMC passes) on one box; outputs under gpurun_out/final/ like tools/gpu_r4_final.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -q > $O/pytest_train.log 2>&1; tail -1 $O/pytest_train.log
timeout 900 python bench.py --workload train-synth256 --steps 50 --warmup 5 --cpu-seconds 30 > $O/bench_train.log 2>&1
for b in 16 32; do timeout 600 python bench.py --workload train-synth256 --train-batch $b --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_train_b$b.log 2>&1; done
bash tools/gpu_train_prof.sh final/train_prof 8 > $O/train_prof.txt 2>&1
bash tools/gpu_train_timeline.sh final/train_tl 8 > $O/train_tl.txt 2>&1
bash tools/gpu_train_pmc.sh final/train_pmc > $O/train_pmc.txt 2>&1
for f in $O/bench_train*.log; do grep "^{" $f | cut -c1-200; done
