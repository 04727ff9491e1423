"""world_size-2 (3, and 8) gloo runs of the sharded whole-slide path on CPU: the band partition + halo exchange +
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
                                                  (2, "v2_solo_like", 260, 40, 1),   # 6 + 5 patch rows: 4 slabs per band
                                                  # the target topology: 8 ranks, 19 patch rows -> bands of 3/3/3/2/2/2/2/2
                                                  (8, "legacy_k3_x2", 12 * 19 - 5, 30, 2)])
def test_sharded_equals_single_process(tmp_path, world, hp_name, H, W, C):
    from oracle import oracle
    from unmicst_amd import model
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2" if world < 8 else "1")
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


def test_library_band_geometry_equals_the_python_schedule():
    """umx_shard_plan (the geometry umx_infer_image_sharded_dev runs, unmicst_amd/csrc/umx_shard.hip) against
    unmicst_amd.sharding's band_partition / needed_image_rows / owned_rows / slab_rows over a grid of image sizes, world
    sizes and slab counts -- including worlds larger than the number of patch rows."""
    import helpers
    from unmicst_amd import model, sharding, umx
    hps = [helpers.small_hps()["v2_duo_like"], model.KNOWN_HP["nucleiDAPI1-5"], model.KNOWN_HP["synthetic-256"]]
    checked = 0
    for hp in hps:
        m = hp.margin
        sub = hp.imSize - 2 * m
        for H in (5, sub, sub + 1, 3 * sub - 1, 7 * sub + 13, 2048, 16384):
            npr = -(-H // sub)
            for world in (1, 2, 3, 8):
                bands = sharding.band_partition(npr, world)
                active = [b - a for a, b in bands if b > a]
                for nslabs in (1, 2, 4):
                    n = max(1, min(nslabs, min(active)))
                    for rank in range(world):
                        pa, pb = bands[rank]
                        for i in range(n):
                            got = umx.shard_plan(hp, H, 77, rank, world, nslabs, i)
                            assert (got["patch_row0"], got["patch_row1"]) == (pa, pb)
                            assert (got["need_row0"], got["need_row1"]) == sharding.needed_image_rows(pa, pb, sub, m, hp.imSize, H)
                            assert (got["own_row0"], got["own_row1"]) == sharding.owned_rows(pa, pb, npr, sub, m, H)
                            assert (got["slab_row0"], got["slab_row1"]) == sharding.slab_rows(pa, pb, npr, sub, m, H, n, i)
                            assert got["nslabs"] == n
                            checked += 1
    assert checked > 500
