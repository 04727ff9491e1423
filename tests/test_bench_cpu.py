"""bench.py's launch logic on CPU: `--gpus N` outside torch.distributed.run starts the N ranks as a child process tree
(before anything touches a GPU) and relays their output and exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_gpus_2_launches_two_ranks():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--master-port", str(_free_port())],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    import re
    seen = sorted(re.findall(r"bench\.py dry launch: rank (\d) of 2 \(local rank (\d)\)", r.stdout))   # (lines may interleave)
    assert seen == [("0", "0"), ("1", "1")], r.stdout[-500:]
    # the "ranks" block of the N > 1 line, filled from the communicator the two ranks formed (gloo here, RCCL on the GPUs)
    import json
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-800:]
    blk = json.loads(lines[0])["ranks"]
    assert blk["world_size_from_communicator"] == 2 and blk["backend"] == "gloo"
    # default workload, strong scaling (round 4): the north star's 16384 x 16384 slide, 86 x 86 sub-patches of 192 pixels, split into
    # two bands of 43 patch rows
    assert blk["patch_rows"] == 86 and blk["patch_cols"] == 86
    per = blk["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    assert per[0]["patch_row0"] == 0 and per[0]["patch_row1"] == per[1]["patch_row0"] == 43 and per[1]["patch_row1"] == 86
    assert sum(p["tiles"] for p in per) == blk["tiles_total"] == 86 * 86
    assert per[0]["owned_image_rows"][0] == 0 and per[0]["owned_image_rows"][1] == per[1]["owned_image_rows"][0]
    assert per[1]["owned_image_rows"][1] == 16384
    ag = blk["allgather_bytes_per_step"]
    assert sum(ag["contributed_per_rank"]) == ag["received_per_rank"] == 3 * 16384 * 16384 and ag["element"] == "uint8"   # (uint8 slabs since round 5)


def test_bench_rejects_a_world_that_does_not_match_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_bench_strong_and_weak_slide_sizes():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "4", "--scaling", "strong"])
    assert a.scaling == "strong" and a.gpus == 4
    # defaults: the north star's slide at every N for the 16384-wide workloads, the fixed-size parity configs as they are
    assert bench.parse_args([]).scaling == "strong" and bench.parse_args(["--workload", "solo-16384"]).scaling == "strong"
    assert bench.parse_args(["--workload", "duo-4096"]).scaling == "weak" and bench.parse_args(["--scaling", "weak"]).scaling == "weak"
    assert bench.WORKLOADS["solo-16384"][0] == "nucleiDAPI1-5"
