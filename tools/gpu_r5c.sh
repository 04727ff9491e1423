cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sharded" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --force-sharded > $O/bench_forced.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > $O/bench_plain.log 2>&1
python - <<'PY'
import json
for f in ("bench_forced","bench_plain"):
    for l in open("gpurun_out/r5c/%s.log"%f, errors="replace"):
        if l.startswith("{"):
            j=json.loads(l)
            print(f, j["value"], j["ms_per_step"], "resident", j["resident"]["value"], j["config"]["checksum"], j["config"]["host_path_equals_resident_path"])
            print("  f32", json.dumps(j.get("f32"))[:600]); print("  train", json.dumps(j.get("train"))[:600]); print("  host_sync", json.dumps(j.get("host_sync"))[:400]); print("  cpu", json.dumps(j.get("cpu_baseline"))[:300])
PY
tail -3 $O/bench_plain.log | cut -c1-300
