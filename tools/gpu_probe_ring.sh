# usage: bash tools/gpu_probe_ring.sh <outdir-name>  -- builds and runs tools/probes/mfma_loops.hip (bare trip, stage skeleton, ring skeleton)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -w tools/probes/mfma_loops.hip -o /tmp/mfma_loops > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
timeout 300 /tmp/mfma_loops > $O/probe_mfma_loops.txt 2>&1
cat $O/probe_mfma_loops.txt
