# usage: bash tools/gpu_power.sh <outdir-name>: sample rocm-smi power / clocks while the resident bench runs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower > $O/smi_idle.txt 2>&1
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi_load.txt 2>&1 &
SMI=$!
python bench.py --steps 200 --warmup 5 --cpu-seconds 0 --resident-only > $O/bench.log 2>&1
kill $SMI 2>/dev/null
grep -E "Power|Max" $O/smi_idle.txt | head -5
sort $O/smi_load.txt | uniq -c | sort -rn | head -12
grep '^{' $O/bench.log | cut -c1-160
