"""world_size-2 (and 3) gloo runs of the sharded whole-slide path on CPU: the band partition + halo exchange +
all-gather host logic must reproduce the single-process oracle result bit for bit on every rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,hp_name,H,W,C", [(2, "v2_duo_like", 100, 61, 2), (3, "legacy_k3_x0", 130, 40, 1),
                                                  (2, "v2_duo_like", 20, 30, 2),    # fewer patch rows than ranks
                                                  (2, "v2_solo_like", 260, 40, 1)])  # 6 + 5 patch rows: 4 slabs per band
def test_sharded_equals_single_process(tmp_path, world, hp_name, H, W, C):
    from oracle import oracle
    from unmicst_amd import model
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "sharded_worker.py"), out, hp_name, str(H), str(W), str(C)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    hp = helpers.small_hps()[hp_name]
    blob = model.random_blob(hp, seed=4)
    img = np.random.default_rng(2).random((C, H, W)) * 0.5
    ref = np.stack([np.array(oracle.single_image_inference(hp, blob, img if C > 1 else img[0], 0.2, 0.2, "accumulate", k))
                    for k in range(hp.nClasses)])
    for rank in range(world):
        got = np.load("%s.rank%d.npy" % (out, rank))
        assert got.dtype == np.float16 and got.shape == ref.shape
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), rank
