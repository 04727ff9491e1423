#!/bin/bash
# usage: tools/kres.sh <file.hip> [grep-pattern]  -- compact per-kernel resource table (VGPRs, SGPRs, spills, occupancy)
cd "$(dirname "$0")/../unmicst_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, subprocess, sys
rows, cur = [], {}
for l in sys.stdin:
    if " error" in l: print(l.rstrip())
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]): (\S+)", l)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"^void umx::", "", n); n = re.sub(r"\(umx::.*", "", n)
    print("%-60s vgpr %-4s sgpr %-4s spill s/v %s/%s occ %s" % (n[:60], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]")))
' | grep -E "${2:-.}"
