# usage: bash tools/gpu_f6.sh <outdir-name>  -- parity report of the fp6 cross-term form, then the resident bench A/B (f16x3 vs f16f6) with per-layer tables
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
UMX_DEBUG_PLAN=1 timeout 900 python tests/f16f6_parity_report.py > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
grep -v "^W2026\|^E2026\|amdgpu.ids\|umx plan" $O/parity.log | tail -12
grep "fp6-cross" $O/parity.log | sort | uniq | tail -12
for prec in f16x3 f16f6 f16x3 f16f6; do
  timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --resident-only --scaling weak --breakdown --precision $prec > $O/bench_$prec.log 2>&1
  python3 - $O/bench_$prec.log $prec <<'PY'
import json, sys
line = None
for l in open(sys.argv[1], errors="replace"):
    if l.startswith("{"): line = l
if line is None: print(sys.argv[2], "FAILED"); sys.exit(0)
j = json.loads(line); r = j.get("roofline") or {}
print("%-8s %9.1f tiles/s %8.3f ms/step  dom frac %.4f" % (sys.argv[2], j["value"], j["ms_per_step"], r.get("frac", 0)))
PY
done
grep -v "^W2026\|^E2026\|amdgpu.ids\|^{" $O/bench_f16f6.log | tail -20
