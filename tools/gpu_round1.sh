set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1b
python -m pytest tests -m gpu -x -q > gpurun_out/r1b/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1b/pytest_gpu.log
python bench.py --steps 5 --warmup 1 --breakdown > gpurun_out/r1b/bench_default.log 2>&1
for b in 32 128 256; do python bench.py --steps 3 --warmup 1 --batch $b --cpu-seconds 0 --breakdown > gpurun_out/r1b/bench_b$b.log 2>&1; done
rocprofv3 --kernel-trace --stats -d gpurun_out/r1b/stats -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/r1b/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r1b/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > gpurun_out/r1b/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r1b/pmc_write -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > gpurun_out/r1b/prof_write.log 2>&1
ls -la gpurun_out/r1b/*
nproc; rocminfo | grep -i -m3 "compute unit\|gfx"
