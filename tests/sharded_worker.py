"""Worker for tests/test_sharding_cpu.py: one rank of a gloo world running unmicst_amd.sharding on CPU tensors.

The engine here is an ORACLE-backed stand-in with the same method surface as umx.Engine (tests may use the oracle;
the product never does): it lets the band partition / halo exchange / all-gather host logic run without a GPU.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OracleEngine:
    def __init__(self, hp, blob):
        self.hp, self.blob = hp, blob

    def tile_grid(self, H, W):
        m = int(self.hp.imSize / 8)
        sub = self.hp.imSize - 2 * m
        npr, npc = -(-H // sub), -(-W // sub)
        return npr, npc, npr * sub + 2 * m, npc * sub + 2 * m

    def synchronize(self):
        pass

    def band_tiles_dev(self, image_ptr, C, H, W, band_row0, band_rows, mean, std, pr0, pr1, probs_ptr):
        from oracle import oracle, pi2d_oracle
        hp = self.hp
        band = _arr(image_ptr, (C, band_rows, W), np.float64)
        full = np.full((C, H, W), np.nan)            # rows outside the band must never be read
        full[:, band_row0:band_row0 + band_rows] = band
        npr, npc, _, _ = self.tile_grid(H, W)
        pi = pi2d_oracle.PI2DOracle(full if C > 1 else full[0], hp.imSize, int(hp.imSize / 8), "accumulate")
        n = (pr1 - pr0) * npc
        x = pi2d_oracle.normalised_batch(pi, pr0 * npc, n, hp.nChannels, mean, std, C == 1)
        assert np.isfinite(x).all(), "band does not cover the rows its patch rows need"
        out = _arr(probs_ptr, (n, hp.imSize, hp.imSize, hp.nClasses), np.float32)
        out[...] = oracle.forward(hp, self.blob, x)

    def stitch_dev(self, probs_ptr, tpr0, tpr1, H, W, mode, stitch, y0, y1, out_ptr):
        from oracle import pi2d_oracle
        hp = self.hp
        assert stitch == 0
        npr, npc, _, _ = self.tile_grid(H, W)
        probs = _arr(probs_ptr, ((tpr1 - tpr0) * npc, hp.imSize, hp.imSize, hp.nClasses), np.float32)
        out = _arr(out_ptr, (hp.nClasses, y1 - y0, W), np.float16)
        for k in range(hp.nClasses):
            pi = pi2d_oracle.PI2DOracle(np.zeros((H, W)), hp.imSize, int(hp.imSize / 8),
                                        "replace" if mode == 1 else "accumulate")
            pi.create_output(1)
            for t in range(tpr0 * npc, tpr1 * npc):
                pi.patch_output(t, probs[t - tpr0 * npc, :, :, k])
            out[k] = np.array(pi.get_valid_output())[y0:y1]


def main():
    import torch
    import torch.distributed as dist
    import helpers
    from unmicst_amd import model, sharding
    out_path, hp_name, H, W, C = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hp = helpers.small_hps()[hp_name]
    blob = model.random_blob(hp, seed=4)
    img = np.random.default_rng(2).random((C, H, W)) * 0.5
    eng = OracleEngine(hp, blob)
    npr, npc, _, _ = eng.tile_grid(H, W)
    m = int(hp.imSize / 8)
    sub = hp.imSize - 2 * m
    pa, pb = sharding.band_partition(npr, world)[rank]
    r0, r1 = sharding.needed_image_rows(pa, pb, sub, m, hp.imSize, H)
    band = torch.from_numpy(np.ascontiguousarray(img[:, r0:max(r1, r0 + 1)]))   # each rank holds ONLY its rows
    full = sharding.infer_image_sharded(eng, band, r0, H, W, 0.2, 0.2, 0, 0, nslabs=4)   # clamped to the smallest band
    np.save("%s.rank%d.npy" % (out_path, rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
