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
    assert bench.WORKLOADS["solo-16384"][0] == "nucleiDAPI1-5"
