# usage: bash tools/gpu_probe.sh <outdir-name> <probe name under tools/probes, without .hip> [more probe names]  -- build and run probes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
for p in "$@"; do
  hipcc -O3 --offload-arch=gfx950 -w tools/probes/$p.hip -o /tmp/$p > $O/build_$p.log 2>&1 || { tail $O/build_$p.log; continue; }
  timeout 300 /tmp/$p > $O/probe_$p.txt 2>&1
  echo "== $p"; cat $O/probe_$p.txt
done
