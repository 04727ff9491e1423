# usage: bash tools/gpu_train_ab.sh <outdir> <variant-file> -- same-box A/B of the training bench (images/s, ms/step, phases)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
while IFS='|' read -r label envs bargs; do
  label=$(echo $label); [ -z "$label" ] && continue
  env $envs timeout 600 python bench.py --cpu-seconds 0 $bargs > $O/$label.log 2>&1
  python3 - "$O/$label.log" "$label" <<'PY'
import json, sys
line = None
for l in open(sys.argv[1], errors="replace"):
    if l.startswith("{"):
        line = l
if line is None:
    print("%-20s FAILED" % sys.argv[2]); sys.exit(0)
j = json.loads(line)
print("%-20s %9.1f images/s %8.3f ms/step  %s  loss %.6f -> %.6f" % (sys.argv[2], j["value"], j["ms_per_step"], j["config"]["phase_ms_per_step"], j["config"]["loss_first"], j["config"]["loss_last"]))
PY
done < $2
