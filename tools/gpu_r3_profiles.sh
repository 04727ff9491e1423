# usage: bash tools/gpu_r3_profiles.sh  -- per-layer rocprofv3 summaries (kernel stats + separate PMC passes) of the headline
# workload and of the two shipped model shapes; outputs under gpurun_out/prof_<workload>/
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh prof_synth256 > gpurun_out/prof_synth256.log 2>&1
UMX_PMC_NAMES=pi2d.gather_normalise,ld0.conv,ld1.conv,ld2.conv,ld3.conv,lb.conv,lu3.convT,lu3.conv,lu2.convT,lu2.conv,lu1.convT,lu1.conv,lu0.convT,lu0.conv,pi2d.stitch \
  bash tools/gpu_pmc.sh prof_solo16384 --workload solo-16384 > gpurun_out/prof_solo16384.log 2>&1
bash tools/gpu_pmc.sh prof_duo4096 --workload duo-4096 > gpurun_out/prof_duo4096.log 2>&1
for w in synth256 solo16384 duo4096; do echo == $w; tail -3 gpurun_out/prof_$w.log | cut -c1-200; ls gpurun_out/prof_$w | tr '\n' ' '; echo; done
timeout 300 python bench.py --workload solo-16384 --steps 5 --warmup 2 --cpu-seconds 0 --breakdown > gpurun_out/bench_solo16384.log 2>&1
timeout 300 python bench.py --workload duo-4096 --steps 5 --warmup 2 --cpu-seconds 0 --breakdown > gpurun_out/bench_duo4096.log 2>&1
