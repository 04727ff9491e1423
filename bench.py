#!/usr/bin/env python3
"""bench.py -- whole-slide tiled inference throughput (tiles/s) on N MI355X of one node.

A "step" is one pass of the hot path over one synthetic slide already resident in HBM:
  PI2D gather + normalise -> UNet forward (all classes) -> fp16-compat stitch  [-> halo exchange + RCCL all-gather, N>1].
Default workload (``wsi-synth256``): BASELINE.json's metric tile (256x256x2, v2 graph, duo widths 36..1152, seeded
weights) on a 2-channel synthetic slide of 2048*N x 16384 px -- each rank holds a 2048-row band, so per-GPU work is
fixed ("weak") and N=8 is exactly the 16384 x 16384 slide of the north star (86 x 86 = 7396 tiles).
Other workloads (parity-test configs of BASELINE.json, not the headline): solo-1024, duo-4096, legacy-105.

Prints ONE JSON line on rank 0.  Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: bf16/f16 MFMA dense peak (~2.5 PF; 16x the fp32 matrix rate)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (model key, channels in image, band rows per GPU, cols)
    "wsi-synth256": ("synthetic-256", 2, 2048, 16384),
    "solo-1024": ("nucleiDAPI1-5", 1, 1024, 1024),
    "duo-4096": ("nucleiDAPILAMIN", 2, 4096, 4096),
    "legacy-1024": ("nucleiDAPI", 1, 1024, 1024),
}


def synth_rows(torch, C, row0, rows, W, device):
    """Deterministic synthetic slide content for image rows [row0, row0+rows): smooth structure + hashed noise in
    [0, 0.983] (what the driver's rescale_intensity produces), float64 [C, rows, W].  Any rank can generate any rows."""
    y = torch.arange(row0, row0 + rows, device=device, dtype=torch.float64)[None, :, None]
    x = torch.arange(W, device=device, dtype=torch.float64)[None, None, :]
    c = torch.arange(C, device=device, dtype=torch.float64)[:, None, None]
    base = 0.22 + 0.18 * torch.sin(y / 37.0 + c) * torch.cos(x / 53.0) + 0.1 * torch.sin((x + 2 * y) / 11.0 + 2 * c)
    h = torch.frac(torch.sin(x * 12.9898 + y * 78.233 + c * 37.719) * 43758.5453)
    return torch.clamp(base + 0.12 * h, 0.0, 0.983).contiguous()


def pmc_traffic(kernel, precision, batch):
    """HBM bytes per launch of `kernel` (a template instantiation as rocprofv3 names it, e.g. "conv_f16x3<9, 4, 1>"),
    averaged over its launches, from the committed rocprofv3 counter passes of this same command
    (profiles/r01/final_<precision>_b<batch>_by_layer_pmc.csv, written by tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE collected
    in separate --pmc passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  None when no profile of
    this configuration is committed."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01", "final_%s_b%d_by_layer_pmc.csv" % (precision, batch))
    if not os.path.exists(path):
        return None
    calls = rd = wr = us = 0.0
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if kernel in r["kernel"] and r.get("hbm_read_MB_corrected") and r.get("hbm_write_MB"):
                n = float(r["calls"])
                calls += n
                rd += n * float(r["hbm_read_MB_corrected"]) * 1e6
                wr += n * float(r["hbm_write_MB"]) * 1e6
                us += n * float(r["avg_us"])
    if calls == 0:
        return None
    return {"bytes_per_launch": round((rd + wr) / calls), "read": round(rd / calls), "write": round(wr / calls),
            "avg_launch_us_profiled": round(us / calls, 2), "launches_profiled": int(calls),
            "source": os.path.relpath(path, ROOT)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wsi-synth256", choices=sorted(WORKLOADS) + ["train-synth256"])
    ap.add_argument("--train-batch", type=int, default=8, help="train-synth256: images per optimisation step")
    ap.add_argument("--batch", type=int, default=256, help="tiles per UNet launch group")
    ap.add_argument("--precision", default="default", choices=["default", "f32", "f16x3"],
                    help="conv arithmetic: exact fp32 MFMA, or 3 binary16 MFMA products per fp32 product (default)")
    ap.add_argument("--band-rows", type=int, default=0, help="override rows per GPU")
    ap.add_argument("--cols", type=int, default=0, help="override slide width")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--slabs", type=int, default=2, help="N>1 path: row slabs per band (stitch + async all-gather each)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (band halo exchange + slab all-gathers) even in a world of one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from unmicst_amd import model, sharding, umx

    if args.workload == "train-synth256":
        return bench_train(args, torch)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a MI355X (there is no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    key, C_img, band_rows, W = WORKLOADS[args.workload]
    band_rows = args.band_rows or band_rows
    W = args.cols or W
    hp = model.KNOWN_HP[key]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = {"synthetic-256": (0.18, 0.17), "nucleiDAPI1-5": (0.34, 0.25), "nucleiDAPILAMIN": (0.18, 0.17),
                 "nucleiDAPI": (0.19808, 0.16236)}[key]
    H = band_rows * world

    eng = umx.Engine(hp, blob, device=local_rank, max_batch=args.batch, precision=args.precision)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    npr, npc, _, _ = eng.tile_grid(H, W)
    tiles_total = npr * npc
    margin = hp.margin
    sub = hp.imSize - 2 * margin
    pa, pb = sharding.band_partition(npr, world)[rank]
    r0, r1 = sharding.needed_image_rows(pa, pb, sub, margin, hp.imSize, H)
    band = synth_rows(torch, C_img, r0, max(r1 - r0, 1), W, dev)   # resident in HBM before the timed region
    out_full = torch.empty((hp.nClasses, H, W), dtype=torch.float16, device=dev) if not sharded else None

    def step():
        if not sharded:
            eng.infer_image_dev(band.data_ptr(), C_img, H, W, mean, std, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT,
                                out_full.data_ptr())
            return out_full
        return sharding.infer_image_sharded(eng, band, r0, H, W, mean, std, umx.MODE_ACCUMULATE,
                                            umx.STITCH_FP16_COMPAT, nslabs=args.slabs)

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    checksum = float(res.float().mean().item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = tiles_total * args.steps / elapsed
        # ---- roofline of the dominant kernel = the template instantiation with the largest share of the timed region (what
        # `rocprofv3 --stats` ranks first: profiles/r01/final_*_kernel_stats.csv), from the in-library HIP events; one kernel
        # serves several layers, so the figures are launch-weighted averages over its sites
        convs = [p for p in prof if p["kernel"].startswith("conv_")]
        by_kernel = {}
        for p in convs:
            k = by_kernel.setdefault(p["kernel"], {"ms": 0.0, "flops": 0.0, "exec": 0.0, "bytes": 0.0, "launches": 0, "layers": []})
            k["ms"] += p["total_ms"]; k["flops"] += p["flops"]; k["exec"] += p["exec_flops"]; k["bytes"] += p["bytes"]
            k["launches"] += p["launches"]; k["layers"].append(p["name"])
        dom_name = max(by_kernel, key=lambda n: by_kernel[n]["ms"])
        dom = by_kernel[dom_name]
        dom_tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        slow = max(convs, key=lambda p: p["total_ms"])          # the single most expensive layer, for the record
        all_flops = sum(p["flops"] for p in convs)
        all_exec = sum(p["exec_flops"] for p in convs)
        all_ms = sum(p["total_ms"] for p in convs)
        peak = PEAK_F16_MFMA_TFLOPS if eng.precision == "f16x3" else PEAK_F32_MFMA_TFLOPS
        roofline = {
            # achieved = ALGORITHMIC fp32 FLOPs of the kernel's launches / HIP-event time of those launches; peak = dense MFMA
            # peak of the dtype the matrix cores run (f16x3 issues 3 binary16 MFMA FLOPs per algorithmic FLOP: "mfma_issued")
            "bound": "mfma", "achieved": round(dom_tflops, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(dom_tflops / peak, 4), "traffic": pmc_traffic(dom_name, eng.precision, args.batch),
            "kernel": dom_name, "layers": dom["layers"], "share_of_step": round(dom["ms"] / (1e3 * elapsed), 4),
            "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
            "flop_per_launch": dom["flops"] / dom["launches"],
            "mfma_issued": {"tflops": round(dom["exec"] / (dom["ms"] * 1e-3) / 1e12, 2),
                            "frac": round(dom["exec"] / (dom["ms"] * 1e-3) / 1e12 / peak, 4)},
            "compulsory_hbm": {"GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                               "frac_of_8TBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
            "slowest_layer": {"layer": slow["name"], "kernel": slow["kernel"],
                              "achieved": round(slow["flops"] / (slow["total_ms"] * 1e-3) / 1e12, 2),
                              "frac": round(slow["flops"] / (slow["total_ms"] * 1e-3) / 1e12 / peak, 4),
                              "avg_launch_us": round(1e3 * slow["total_ms"] / slow["launches"], 2)},
            "all_conv_launches": {"achieved": round(all_flops / (all_ms * 1e-3) / 1e12, 2),
                                  "frac": round(all_flops / (all_ms * 1e-3) / 1e12 / peak, 4),
                                  "mfma_issued_frac": round(all_exec / (all_ms * 1e-3) / 1e12 / peak, 4),
                                  "share_of_step": round(all_ms / (1e3 * elapsed), 4)},
        }
        if args.breakdown:
            print("%-24s %-30s %8s %10s %9s %9s" % ("layer", "kernel", "launches", "total_ms", "TFLOP/s", "GB/s"),
                  file=sys.stderr)
            for p in sorted(prof, key=lambda p: -p["total_ms"]):
                s = p["total_ms"] * 1e-3
                print("%-24s %-30s %8d %10.3f %9.2f %9.1f" % (p["name"], p["kernel"], p["launches"], p["total_ms"],
                                                              p["flops"] / s / 1e12, p["bytes"] / s / 1e9),
                      file=sys.stderr)
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(hp, blob, band, mean, std, args.cpu_seconds)
        line = {
            "metric": "tiles/sec (%dx%dx%d) whole-slide inference" % (hp.imSize, hp.imSize, hp.nChannels),
            "value": round(value, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16x3 (fp32 products as 3 binary16 MFMA products, fp32 accumulate)"}[
                eng.precision], "data": "synthetic",
            "config": {"workload": "%s: %s hp (%s graph, seeded weights), %d-channel synthetic slide %dx%d, "
                                   "%d tiles/step, batch %d, fp16-compat stitch%s" % (
                                       args.workload, key, "v2" if hp.graph else "legacy", C_img, H, W, tiles_total,
                                       args.batch, ", band halo exchange + slab-wise RCCL all-gather" if sharded else ""),
                       "tiles_per_step": tiles_total, "slide": [H, W], "flop_per_tile_as_written": hp.flops_per_tile(),
                       "flop_per_tile_executed_unpadded": umx.describe(hp)["flops_per_tile"], "checksum": checksum},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    eng.close()
    if sharded:
        dist.destroy_process_group()


def bench_train(args, torch):
    """BASELINE.json configs[4]: forward + backward + Adam on random 256x256x2 batches (synthetic-256 hp, the duo
    script's regime), 1 x MI355X.  A step = one umx_train_step on device-resident batches.  Not the headline metric:
    select it with --workload train-synth256.  N > 1 would need a gradient all-reduce the reference does not have
    (it trains on one GPU): replicas only, not run here."""
    from unmicst_amd import model, trainer
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        print("bench.py: the training workload runs on one GPU (the reference trains on one)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a MI355X (there is no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    dev = torch.device("cuda", 0)
    hp = model.KNOWN_HP["synthetic-256"]
    blob = model.random_blob(hp, seed=20260101)
    B = args.train_batch
    opts = trainer.duo_options()
    tr = trainer.Trainer(hp, blob, opts, batch=B, device=0)
    g = torch.Generator(device="cpu").manual_seed(20260101)
    nb = 4                                       # distinct batches, cycled
    data = torch.randn((nb, B, hp.imSize, hp.imSize, hp.nChannels), generator=g).to(dev)
    cls = torch.randint(0, hp.nClasses, (nb, B, hp.imSize, hp.imSize), generator=g)
    labels = torch.nn.functional.one_hot(cls, hp.nClasses).float().to(dev)
    weights = (0.5 + 2.5 * torch.rand((nb, B, hp.imSize, hp.imSize, hp.nClasses), generator=g)).to(dev)
    torch.cuda.synchronize(dev)

    def step(i):
        j = i % nb
        tr.step_dev(data[j].data_ptr(), labels[j].data_ptr(), weights[j].data_ptr())

    for i in range(args.warmup):
        step(i)
    first = tr.loss()[0]
    tr.profile(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    last = tr.loss()[0]                          # synchronises the trainer's stream
    elapsed = time.perf_counter() - t0
    ph = tr.profile(False)
    flops_step = tr.flops_per_image * B
    tflops = flops_step * args.steps / elapsed / 1e12
    cpu = None
    if args.cpu_seconds > 0:
        cpu = cpu_baseline_train(hp, blob, data[0].cpu().numpy(), labels[0].cpu().numpy(), weights[0].cpu().numpy(),
                                 args.cpu_seconds)
    line = {
        "metric": "training images/sec (%dx%dx%d) forward+backward+Adam" % (hp.imSize, hp.imSize, hp.nChannels),
        "value": round(B * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "train-synth256: synthetic-256 hp (v2 graph, seeded weights), batch %d, duo regime "
                               "(Adam 6e-5, L2, dropout), random 256x256x2 batches resident in HBM" % B,
                   "batch": B, "flop_per_image": tr.flops_per_image, "loss_first": first, "loss_last": last,
                   "phase_ms_per_step": {k: round(v / max(ph["steps"], 1), 3) for k, v in ph.items() if k != "steps"}},
        # whole step against the fp32 matrix peak: forward + input-gradient + weight-gradient convolutions are
        # ~all of the algorithmic FLOPs; the element-wise BN/activation passes are HBM-bound and show up as lost fraction
        "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None, "kernel": "whole training step",
                     "flop_per_launch": flops_step},
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    tr.close()


def cpu_baseline_train(hp, blob, data, labels, weights, budget_s):
    """The training oracle (torch autograd on the host cores, float32) on the first images of the same batch."""
    import torch
    from oracle import train_oracle as to
    o = to.duo_options()
    t = time.perf_counter()
    to.loss_and_grads(hp, blob, data[:1], labels[:1], weights[:1], o, 0, dtype=torch.float32)   # untimed: thread pool, caches
    warm = time.perf_counter() - t
    n = int(max(1, min(data.shape[0], budget_s / max(warm, 1e-3) / 2)))
    t = time.perf_counter()
    to.loss_and_grads(hp, blob, data[:n], labels[:n], weights[:n], o, 0, dtype=torch.float32)
    dt = time.perf_counter() - t
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "forward+backward of %d image(s) of the same batch (oracle/train_oracle.py, torch CPU float32, no "
                      "optimiser update), %.1f s" % (n, dt)}


def cpu_baseline(hp, blob, band, mean, std, budget_s):
    """The oracle (C restatement, OpenMP on every host core) on a bounded sample of the same workload's tiles."""
    from oracle import oracle, pi2d_oracle
    P, m = hp.imSize, hp.margin
    sub = P - 2 * m
    rows = min(band.shape[1], 2 * sub + 2 * m)
    cols = min(band.shape[2], 8 * sub + 2 * m)
    crop = band[:, :rows, :cols].cpu().numpy()
    if hp.nChannels == 1:
        crop = crop[0]
    pi = pi2d_oracle.PI2DOracle(crop, P, m, "accumulate")
    x1 = pi2d_oracle.normalised_batch(pi, 0, 1, hp.nChannels, mean, std, False)
    t = time.perf_counter()
    oracle.forward(hp, blob, x1)
    per_tile = max(time.perf_counter() - t, 1e-4)
    n = int(max(1, min(pi.num_patches, budget_s / per_tile)))
    t = time.perf_counter()
    done = 0
    while done < n:
        nb = min(4, n - done)
        oracle.forward(hp, blob, pi2d_oracle.normalised_batch(pi, done, nb, hp.nChannels, mean, std, False))
        done += nb
    dt = time.perf_counter() - t
    return {"value": round(n / dt, 3), "unit": "tiles/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "first %d tiles of the same slide (PI2D gather+normalise + UNet forward, oracle/unet_oracle.c, "
                      "double-accumulate fp32, OpenMP), %.1f s" % (n, dt)}


if __name__ == "__main__":
    main()
