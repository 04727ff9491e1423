#!/usr/bin/env python3
"""What ONE rank of an N-rank world spends on its band of the bench slide, measured on one GPU: the rank's whole schedule of
umx_infer_image_sharded_raw_submit (staged upload, launch groups, stitch, uint8 cast, own-rows download, two slides in flight) with
the inter-rank operations replaced by no-ops behind umx_shard_init_transport (send / recv do nothing: the halo row a rank receives
is garbage, its arithmetic cost is the same; all-gather copies nothing).  NOT a measurement of RCCL or xGMI -- it prices the
compute side of the N > 1 line, i.e. what the scaling curve can at best look like (max over the ranks' times), and it is how the
launch-group schedule of umx_shard.hip was chosen.   usage: python tools/shard_rank_timing.py [--world 8] [--steps 10]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--slabs", type=int, default=2)
    a = ap.parse_args()
    import torch
    import bench
    from unmicst_amd import model, umx
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hp = model.KNOWN_HP["synthetic-256"]
    blob = model.random_blob(hp, seed=20260101)
    H = W = 16384
    mean, std = bench.NORMALISATION["synthetic-256"]
    K = hp.nClasses
    out = {"world": a.world, "slide": [H, W], "ranks": []}
    full = torch.empty((K, H, W), dtype=torch.uint8, device=dev)
    for rank in sorted({0, a.world // 2, a.world - 1}):
        with umx.Engine(hp, blob, device=0, max_batch=umx.auto_batch(hp)) as eng:
            eng.shard_init_transport(lambda *x: None, lambda *x: None, lambda *x: None, rank, a.world,
                                     group_start=lambda: None, group_end=lambda: None)
            pl = eng.shard_plan(H, W, rank, a.world, a.slabs)
            r0, r1, o0, o1 = pl["need_row0"], pl["need_row1"], pl["own_row0"], pl["own_row1"]
            band = torch.empty((2, r1 - r0, W), dtype=torch.int16).pin_memory()
            band.copy_(bench.synth_rows_u16(torch, 2, r0, r1 - r0, W, dev).to(torch.int16))
            owns = [torch.empty((K, o1 - o0, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
            inflight = []

            def step(i):
                slot = i & 1
                if len(inflight) == 2:
                    eng.infer_image_wait(inflight.pop(0))
                eng.infer_image_sharded_raw_submit(slot, band.data_ptr(), 16, 2, H, W, r0, r1 - r0, None, mean, std, umx.MODE_ACCUMULATE,
                                                   a.slabs, owns[slot].data_ptr(), full.data_ptr())
                inflight.append(slot)

            def drain():
                while inflight:
                    eng.infer_image_wait(inflight.pop(0))
                eng.synchronize()
            for i in range(2):
                step(i)
            drain()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(i)
            drain()
            dt = (time.perf_counter() - t0) / a.steps
            npr, npc, _, _ = eng.tile_grid(H, W)
            tiles = (pl["patch_row1"] - pl["patch_row0"]) * npc
            out["ranks"].append({"rank": rank, "patch_rows": [pl["patch_row0"], pl["patch_row1"]], "tiles": tiles,
                                 "ms_per_step": round(1e3 * dt, 3), "tiles_per_s": round(tiles / dt, 1)})
    worst = max(r["ms_per_step"] for r in out["ranks"])
    out["projected"] = {"tiles_per_s_all_ranks": round(7396 / (worst * 1e-3), 1),
                        "note": "whole-slide tiles / the slowest measured rank's step: compute side only, no inter-rank traffic"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
