# usage: bash tools/gpu_ab.sh <outdir-name> <variant-file>
# Each non-empty line of the variant file: "<label> | <ENV=... ENV=...> | <bench args>"; runs bench.py once per line
# (same box, back to back) and prints label + tiles/s + ms/step.  Logs under gpurun_out/<name>/<label>.log
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
    print("%-28s FAILED" % sys.argv[2]); sys.exit(0)
j = json.loads(line)
r = j.get("roofline") or {}
print("%-28s %9.1f tiles/s %8.3f ms/step  dom %s frac %.4f  checksum %s" % (sys.argv[2], j["value"], j["ms_per_step"], r.get("kernel"), r.get("frac", 0), j["config"].get("checksum")))
PY
done < $2
