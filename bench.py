#!/usr/bin/env python3
"""bench.py -- whole-slide tiled inference throughput (tiles/s) on N MI355X of one node.

A "step" is one pass of the hot path over one synthetic slide, measured the way SURVEY.md section 8(d) scopes it:
  H2D of the slide -> PI2D gather + normalise -> UNet forward (all classes) -> fp16-compat stitch
  [-> band halo exchange + RCCL all-gather, N>1] -> D2H of the probability stack.
N = 1 goes through the library's host entry point (umx_infer_image_raw: uint16 planes up, uint8 planes down, in row slabs on
two copy streams under the tile kernels); the same slide resident in HBM (umx_infer_image_dev) is timed right after it
and reported as `resident` in the same JSON line (kernel-only number, never `value`).
Default workload (``wsi-synth256``): BASELINE.json's metric tile (256x256x2, v2 graph, duo widths 36..1152, seeded
weights) on the north star's 2-channel synthetic 16384 x 16384 slide (86 x 86 = 7396 tiles) at EVERY N ("strong": N ranks split
the same slide into bands of patch rows; the N = 1 line is the denominator of the 1 -> 8 curve, 20 steps = 7 s of timed region).
``--scaling weak`` gives every rank a 2048-row band of a 2048*N x 16384 slide instead (946 tiles per GPU: rounds 1-3's line; at
N = 1 the default line carries it as `weak_band`).  Other workloads are parity-test configs of BASELINE.json, not the headline.

Prints ONE JSON line on rank 0.  Launch: `python bench.py [--gpus N]` -- for N > 1 outside torch.distributed.run this
process starts `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` as a CHILD (before anything
touches the GPU) and relays its output and exit code; under torch.distributed.run it is one rank.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: bf16/f16 MFMA dense peak (~2.5 PF; 16x the fp32 matrix rate)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (model key, channels in image, band rows per GPU, cols)
    "wsi-synth256": ("synthetic-256", 2, 2048, 16384),
    "solo-16384": ("nucleiDAPI1-5", 1, 2048, 16384),     # BASELINE.json configs[3] as worded: solo hp, 16384^2 at N = 8
    "solo-1024": ("nucleiDAPI1-5", 1, 1024, 1024),
    "duo-4096": ("nucleiDAPILAMIN", 2, 4096, 4096),
    "legacy-1024": ("nucleiDAPI", 1, 1024, 1024),
}
NORMALISATION = {"synthetic-256": (0.18, 0.17), "nucleiDAPI1-5": (0.34, 0.25), "nucleiDAPILAMIN": (0.18, 0.17),
                 "nucleiDAPI": (0.19808, 0.16236)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="wsi-synth256", choices=sorted(WORKLOADS) + ["train-synth256"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default for the 16384-wide workloads): the slide of 8 bands (16384 rows: the north star's slide) at "
                         "every N; weak (default for the fixed-size parity configs): every rank holds a band of --band-rows rows "
                         "(the slide grows with N)")
    ap.add_argument("--train-batch", type=int, default=8, help="train-synth256: images per optimisation step")
    ap.add_argument("--batch", type=int, default=0, help="tiles per UNet launch group (0: umx.auto_batch -- 2^24 pixels per group, "
                                                      "i.e. 256 tiles of the 256-pixel metric tile)")
    ap.add_argument("--precision", default="default", choices=["default", "f32", "f16x3", "f16f6"],
                    help="conv arithmetic: exact fp32 MFMA, or 3 binary16 MFMA products per fp32 product (default)")
    ap.add_argument("--band-rows", type=int, default=0, help="override rows per GPU")
    ap.add_argument("--cols", type=int, default=0, help="override slide width")
    ap.add_argument("--cpu-seconds", type=float, default=45.0, help="budget of the CPU-baseline legs (0 = skip)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--profile-every", type=int, default=5,
                    help="bracket every N-th launch of each layer with HIP events inside the timed region (1: every launch, "
                         "costs the synthetic-256 step 1.1 %%; 5 is coprime with the 29 / 4 launch groups of the default slide / band, "
                         "so the sampled launches walk through every group position)")
    ap.add_argument("--slabs", type=int, default=2, help="N>1 path: row slabs per band (stitch + async all-gather each)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (band halo exchange + slab all-gathers) even in a world of one rank")
    ap.add_argument("--native-shard", action="store_true", help="(the only N>1 path; kept for old command lines)")
    ap.add_argument("--resident-only", action="store_true",
                    help="time only the HBM-resident slide (kernel-only; the line's value is then NOT the section-8(d) metric)")
    ap.add_argument("--no-legs", action="store_true", help="skip the exact-fp32 and training legs of the default N = 1 line")
    ap.add_argument("--master-port", type=int, default=29577)
    ap.add_argument("--dry-launch", action="store_true",
                    help="N>1 launch test: every rank prints its rank/world line and exits before touching a GPU")
    args = ap.parse_args(argv)
    if args.scaling is None:
        args.scaling = "strong" if args.workload in ("wsi-synth256", "solo-16384") else "weak"
    return args


def self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child process tree (this process never touches
    the GPU -- replacing a GPU-initialised process image takes the box down, and nothing here has imported torch yet)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def synth_rows_u16(torch, C, row0, rows, W, device):
    """Deterministic synthetic slide content for image rows [row0, row0+rows): smooth structure + hashed noise, uint16
    [C, rows, W] (kept as int32 on the device).  Any rank can generate any rows."""
    y = torch.arange(row0, row0 + rows, device=device, dtype=torch.float64)[None, :, None]
    x = torch.arange(W, device=device, dtype=torch.float64)[None, None, :]
    c = torch.arange(C, device=device, dtype=torch.float64)[:, None, None]
    base = 0.22 + 0.18 * torch.sin(y / 37.0 + c) * torch.cos(x / 53.0) + 0.1 * torch.sin((x + 2 * y) / 11.0 + 2 * c)
    h = torch.frac(torch.sin(x * 12.9898 + y * 78.233 + c * 37.719) * 43758.5453)
    v = torch.clamp(base + 0.12 * h, 0.0, 0.983)
    return torch.round(v * 65535.0).to(torch.int32).contiguous()


def im2double(torch, u16_i32):
    """toolbox/imtools.py:42-53 for uint16: np.multiply(I, 1/65535) in float64."""
    return (u16_i32.to(torch.float64) * (1.0 / 65535)).contiguous()


def pmc_traffic(kernel, precision, batch):
    """HBM bytes per launch of `kernel` (a template instantiation as rocprofv3 names it, e.g. "conv_f16x3<9, 4, 1, false, 4>": N-tiles, M-tiles per wave, phases, stamped twin, pixel-index array),
    averaged over its launches, from the committed rocprofv3 counter passes of this same command
    (profiles/rNN/final_<precision>_b<batch>_by_layer_pmc.csv, written by tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE collected
    in separate --pmc passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  None when no profile of
    this configuration is committed."""
    import csv
    path = None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        cand = os.path.join(ROOT, "profiles", rnd, "final_%s_b%d_by_layer_pmc.csv" % (precision, batch))
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None
    commit = None          # the commit the counters were captured at (sidecar written when the CSV was copied into profiles/)
    if os.path.exists(path + ".commit"):
        commit = open(path + ".commit").read().strip()
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
            "source": os.path.relpath(path, ROOT), "captured_at_commit": commit}


def roofline_of(prof, elapsed_s, eng, batch):
    """Roofline of the dominant kernel = the template instantiation with the largest share of the timed region (what
    `rocprofv3 --stats` ranks first), from the in-library HIP events recorded on the stream the kernels are launched on;
    one kernel serves several layers, so the figures are launch-weighted averages over its sites."""
    convs = [p for p in prof if p["kernel"].startswith("conv_")]
    by_kernel = {}
    for p in convs:
        k = by_kernel.setdefault(p["kernel"], {"ms": 0.0, "flops": 0.0, "exec": 0.0, "bytes": 0.0, "launches": 0, "layers": [],
                                               "ms_all": 0.0, "seen": 0})
        k["ms"] += p["total_ms"]; k["flops"] += p["flops"]; k["exec"] += p["exec_flops"]; k["bytes"] += p["bytes"]
        k["launches"] += p["launches"]; k["layers"].append(p["name"])
        # (sampled profiling: `launches` of `seen` launches carry events; a layer's total = its average x every launch)
        k["ms_all"] += p["total_ms"] * p.get("seen", p["launches"]) / max(p["launches"], 1); k["seen"] += p.get("seen", p["launches"])
    dom_name = max(by_kernel, key=lambda n: by_kernel[n]["ms_all"])
    dom = by_kernel[dom_name]
    dom_tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    slow = max(convs, key=lambda p: p["total_ms"])
    all_flops = sum(p["flops"] for p in convs)
    all_exec = sum(p["exec_flops"] for p in convs)
    all_ms = sum(p["total_ms"] for p in convs)
    all_ms_est = sum(k["ms_all"] for k in by_kernel.values())
    peak = PEAK_F16_MFMA_TFLOPS if eng.precision in ("f16x3", "f16f6") else PEAK_F32_MFMA_TFLOPS
    # every instantiation that shares the dominant kernel's <N-tiles, M-tiles, phases> (since round 6 the long-K convolutions of the duo
    # widths run the fp6 cross-term instantiation of the same 9-tile kernel: rocprofv3 lists it as a kernel of its own)
    def first7(n):   # <NT, KMT, NPH, DBG, MAXP, PK, D2S | F6, W2>
        return n[n.index("<") + 1:-1].split(", ")[:7] if n.startswith("conv_f16x3<") else [n]
    fam_names = sorted(n for n in by_kernel if first7(n) == first7(dom_name))
    fam = [by_kernel[n] for n in fam_names]
    fam_ms, fam_fl, fam_all = sum(k["ms"] for k in fam), sum(k["flops"] for k in fam), sum(k["ms_all"] for k in fam)
    same_template = {"kernels": fam_names, "achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 2),
                     "frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / peak, 4), "share_of_step": round(fam_all / (1e3 * elapsed_s), 4)}
    if eng.hp.graph == 0 and eng.precision != "f32":
        # the legacy graph's 16 - 64-channel layers are HBM / LDS-bound, not matrix-bound: its line is quoted on
        # the compulsory activation bytes of the dominant kernel's launches against 8 TB/s; the matrix fraction rides along
        gbps = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBS, 4),
                "traffic": pmc_traffic(dom_name, eng.precision, batch), "kernel": dom_name, "layers": dom["layers"],
                "share_of_step": round(dom["ms_all"] / (1e3 * elapsed_s), 4), "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                "launches_timed": dom["launches"], "launches_in_region": dom["seen"],
                "as_matrix_work": {"achieved_tflops": round(dom_tflops, 2), "frac_of_binary16_peak": round(dom_tflops / peak, 4)},
                "all_conv_launches": {"achieved": round(all_flops / (all_ms * 1e-3) / 1e12, 2),
                                      "frac": round(all_flops / (all_ms * 1e-3) / 1e12 / peak, 4),
                                      "mfma_issued_frac": round(all_exec / (all_ms * 1e-3) / 1e12 / peak, 4),
                                      "share_of_step": round(all_ms_est / (1e3 * elapsed_s), 4)}}
    return {
        # achieved = ALGORITHMIC fp32 FLOPs of the kernel's launches / HIP-event time of those launches; peak = dense MFMA
        # peak of the dtype the matrix cores run (f16x3 issues 3 binary16 MFMA FLOPs per algorithmic FLOP: "mfma_issued")
        "bound": "mfma", "achieved": round(dom_tflops, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(dom_tflops / peak, 4), "traffic": pmc_traffic(dom_name, eng.precision, batch),
        "kernel": dom_name, "layers": dom["layers"], "share_of_step": round(dom["ms_all"] / (1e3 * elapsed_s), 4),
        "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
        "launches_timed": dom["launches"], "launches_in_region": dom["seen"],
        "flop_per_launch": dom["flops"] / dom["launches"],
        "mfma_issued": {"tflops": round(dom["exec"] / (dom["ms"] * 1e-3) / 1e12, 2),
                        "frac": round(dom["exec"] / (dom["ms"] * 1e-3) / 1e12 / peak, 4)},
        "compulsory_hbm": {"GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                           "frac_of_8TBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
        "same_template": same_template,
        "slowest_layer": {"layer": slow["name"], "kernel": slow["kernel"],
                          "achieved": round(slow["flops"] / (slow["total_ms"] * 1e-3) / 1e12, 2),
                          "frac": round(slow["flops"] / (slow["total_ms"] * 1e-3) / 1e12 / peak, 4),
                          "avg_launch_us": round(1e3 * slow["total_ms"] / slow["launches"], 2)},
        "all_conv_launches": {"achieved": round(all_flops / (all_ms * 1e-3) / 1e12, 2),
                              "frac": round(all_flops / (all_ms * 1e-3) / 1e12 / peak, 4),
                              "mfma_issued_frac": round(all_exec / (all_ms * 1e-3) / 1e12 / peak, 4),
                              "share_of_step": round(all_ms_est / (1e3 * elapsed_s), 4)},
    }


def print_breakdown(prof):
    print("%-24s %-30s %8s %10s %9s %9s" % ("layer", "kernel", "launches", "total_ms", "TFLOP/s", "GB/s"), file=sys.stderr)
    for p in sorted(prof, key=lambda p: -p["total_ms"] * p.get("seen", p["launches"]) / max(p["launches"], 1)):
        s = p["total_ms"] * 1e-3
        est = p["total_ms"] * p.get("seen", p["launches"]) / max(p["launches"], 1)   # all launches (sampled profiling: an estimate)
        print("%-24s %-30s %8d %10.3f %9.2f %9.1f" % (p["name"], p["kernel"], p.get("seen", p["launches"]), est,
                                                      p["flops"] / s / 1e12, p["bytes"] / s / 1e9), file=sys.stderr)


def ranks_block(dist, torch, hp, H, W, rank, agree, dev=None):
    """What the communicator saw, for the N > 1 line: world size from the process group, every rank's band of patch rows and
    tile count, the bytes each rank contributes to / receives from the all-gather per step, and whether the host path equalled
    the resident path on EVERY rank.  Works over any backend (the dry launch uses gloo, no GPU)."""
    from unmicst_amd import sharding
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    sub = hp.imSize - 2 * hp.margin
    npr, npc = -(-H // sub), -(-W // sub)
    pa, pb = sharding.band_partition(npr, world)[rank]
    mine = [int(pa), int(pb), int((pb - pa) * npc)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        agree_all = None
        if agree is not None:
            flags = [None] * world
            dist.all_gather_object(flags, bool(agree))
            agree_all = all(flags)
    else:
        gathered, agree_all = [mine], agree
    K = hp.nClasses
    owned = [sharding.owned_rows(a, b, npr, sub, hp.margin, H) for a, b, _ in gathered]
    return {"world_size_from_communicator": int(world), "backend": dist.get_backend() if world > 1 else None,
            "patch_rows": int(npr), "patch_cols": int(npc),
            "per_rank": [{"rank": i, "patch_row0": g[0], "patch_row1": g[1], "tiles": g[2], "owned_image_rows": [int(o[0]), int(o[1])]}
                         for i, (g, o) in enumerate(zip(gathered, owned))],
            "tiles_total": int(sum(g[2] for g in gathered)),
            # (uint8 planes since round 5: the stitched slabs are cast before they are gathered)
            "allgather_bytes_per_step": {"contributed_per_rank": [int(K * max(o[1] - o[0], 0) * W) for o in owned],
                                         "received_per_rank": int(K * H * W), "element": "uint8"},
            "host_path_equals_resident_path_all_ranks": agree_all}


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(self_launch(args))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.dry_launch:
        sys.stdout.write("bench.py dry launch: rank %d of %d (local rank %d)\n" % (rank, world, local_rank))   # one write
        sys.stdout.flush()
        if world > 1:      # the "ranks" block of the JSON line, filled from a (gloo) communicator: no GPU is touched
            import torch.distributed as dist
            from unmicst_amd import model
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(args.master_port))
            dist.init_process_group("gloo", rank=rank, world_size=world)
            key, C_img, band_rows, W = WORKLOADS[args.workload]
            hp = model.KNOWN_HP[key]
            H = (args.band_rows or band_rows) * (8 if args.scaling == "strong" else world)
            block = ranks_block(dist, None, hp, H, args.cols or W, rank, None)
            if rank == 0:
                print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks": block}))
            dist.destroy_process_group()
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from unmicst_amd import model, sharding, umx

    if args.workload == "train-synth256":
        return bench_train(args, torch)
    if not torch.cuda.is_available():
        print("bench.py needs a MI355X (there is no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    key, C_img, band_rows, W = WORKLOADS[args.workload]
    band_rows = args.band_rows or band_rows
    W = args.cols or W
    hp = model.KNOWN_HP[key]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = NORMALISATION[key]
    H = band_rows * (8 if args.scaling == "strong" else world)

    if args.batch <= 0:
        args.batch = umx.auto_batch(hp)
    # the training leg of the default N = 1 line (BASELINE configs[4]) runs FIRST, on an idle chip like `--workload train-synth256`
    # does: its step is a chain of short launches whose time follows the clock, and behind a minute of inference at the package
    # power cap the same leg measured 9 % lower (1 413 vs 1 578 - 1 599 images/s on one box)
    legs = world == 1 and not sharded and args.workload == "wsi-synth256" and not args.no_legs and not args.resident_only
    train = train_leg(torch, dev) if legs else None
    eng = umx.Engine(hp, blob, device=local_rank, max_batch=args.batch, precision=args.precision)
    # every engine launch, torch op and RCCL call of this process is ordered on ONE non-default stream
    work = torch.cuda.Stream(dev)
    eng.set_stream(work.cuda_stream)
    npr, npc, _, _ = eng.tile_grid(H, W)
    tiles_total = npr * npc
    margin = hp.margin
    sub = hp.imSize - 2 * margin
    pa, pb = sharding.band_partition(npr, world)[rank]
    r0, r1 = sharding.needed_image_rows(pa, pb, sub, margin, hp.imSize, H)
    rows = max(r1 - r0, 1)
    # the slide on the HOST, pinned: uint16 planes [C, rows, W], the rows this rank's tiles read
    with torch.cuda.stream(work):
        band_i32 = synth_rows_u16(torch, C_img, r0, rows, W, dev)
        host_u16 = torch.empty((C_img, rows, W), dtype=torch.int16).pin_memory()
        host_u16.copy_(band_i32.to(torch.int16))                      # bit pattern of the uint16 values
        band_f64 = im2double(torch, band_i32)                          # the HBM-resident twin (kernel-only timing)
        del band_i32
    work.synchronize()
    K = hp.nClasses

    if not sharded:
        host_outs = [torch.empty((K, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
        host_out = host_outs[0]
        dev_out = torch.empty((K, H, W), dtype=torch.float16, device=dev)
        inflight = []

        def step_host():
            # section 8(d): H2D + tiles + stitch + D2H of every slide, through the C ABI's host entry points.  Slides are
            # streamed the way a per-file driver loop would: slide i+1 is submitted (its upload starts) while slide i
            # computes, so at most two are in flight; fence() drains the last one inside the timed region.
            slot = step_host.n & 1
            step_host.n += 1
            if len(inflight) == 2:
                eng.infer_image_wait(inflight.pop(0))
            eng.infer_image_raw_submit(slot, host_u16.data_ptr(), 16, C_img, H, W, False, mean, std,
                                       host_outs[slot].data_ptr())
            inflight.append(slot)
        step_host.n = 0

        def step_resident():
            eng.infer_image_dev(band_f64.data_ptr(), C_img, H, W, mean, std, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT,
                                dev_out.data_ptr())
    else:
        # N > 1 (and --force-sharded at N = 1): the SAME product path as the N = 1 line, band by band -- umx_infer_image_sharded_raw_submit:
        # this rank's uint16 rows up from pinned host memory piece by piece under the tiles, im2double in the tile gather, stitched
        # slabs cast to uint8 and all-gathered over RCCL, the rank's own rows down to pinned host memory, two slides in flight on the
        # library's copy streams.  No torch op inside the timed region.  unmicst_amd/sharding.py (the same schedule over
        # torch.distributed) only CHECKS the first slide; a world that cannot run the native path fails loudly instead of timing another one.
        umx.require_torch_runtime("bench.py --gpus N")
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(umx.Engine.shard_unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(idt, 0)
        eng.shard_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
        y0, y1 = sharding.owned_rows(pa, pb, npr, sub, margin, H)
        full_f16 = torch.empty((K, H, W), dtype=torch.float16, device=dev)
        fulls_u8 = [torch.empty((K, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
        host_owns = [torch.empty((K, max(y1 - y0, 0), W), dtype=torch.uint8).pin_memory() for _ in range(2)]
        inflight = []

        def native_dev(band):
            eng.infer_image_sharded_dev(band.data_ptr(), C_img, H, W, r0, band.shape[1], mean, std, umx.MODE_ACCUMULATE,
                                        umx.STITCH_FP16_COMPAT, args.slabs, full_f16.data_ptr())
            return full_f16
        # first slide, outside the timed region: (1) the in-library schedule on the float64 band == sharding.py's over torch.distributed;
        # (2) the raw entry's gathered uint8 stack and own rows == the drivers' uint8 recipe applied to (1)
        with torch.cuda.stream(work):
            a = native_dev(band_f64).clone()
            b = sharding.infer_image_sharded(eng, band_f64, r0, H, W, mean, std, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT,
                                             nslabs=args.slabs, sync=False)
            same_sched = bool(torch.equal(a, b))
            first = (a * 255.0).to(torch.uint8)
            want_u8 = (255.0 * (first.to(torch.float64) * (1.0 / 255))).to(torch.uint8)
            del a, b, first
        work.synchronize()
        eng.infer_image_sharded_raw_submit(0, host_u16.data_ptr(), 16, C_img, H, W, r0, rows if pb > pa else 0, None, mean, std,
                                           umx.MODE_ACCUMULATE, args.slabs, host_owns[0].data_ptr() if y1 > y0 else 0,
                                           fulls_u8[0].data_ptr())
        eng.infer_image_wait(0)
        same_raw = bool(torch.equal(fulls_u8[0], want_u8)) and bool(torch.equal(host_owns[0], want_u8[:, y0:y1].cpu()))
        okv = torch.tensor([1 if (same_sched and same_raw) else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(okv, op=dist.ReduceOp.MIN)
        if int(okv.item()) != 1:
            print("bench.py: rank %d: the native sharded path is not bit-equal to its checkers (schedule vs sharding.py: %s, raw entry vs "
                  "the uint8 recipe: %s) on some rank -- refusing to time anything else" % (rank, same_sched, same_raw), file=sys.stderr)
            sys.exit(4)
        del want_u8
        shard_path = {"path": "umx_infer_image_sharded_raw_submit (RCCL inside libumx; uint16 band up, uint8 slabs gathered, own rows down)",
                      "first_slide_equals_sharding_py_and_uint8_recipe_all_ranks": True, "note": None}

        def step_host():   # H2D of this rank's band + tiles + halo exchange + stitch + uint8 all-gather + D2H of its own rows
            slot = step_host.n & 1
            step_host.n += 1
            if len(inflight) == 2:
                eng.infer_image_wait(inflight.pop(0))
            eng.infer_image_sharded_raw_submit(slot, host_u16.data_ptr(), 16, C_img, H, W, r0, rows if pb > pa else 0, None, mean, std,
                                               umx.MODE_ACCUMULATE, args.slabs, host_owns[slot].data_ptr() if y1 > y0 else 0,
                                               fulls_u8[slot].data_ptr())
            inflight.append(slot)
            return fulls_u8[slot]
        step_host.n = 0

        def step_resident():
            with torch.cuda.stream(work):
                return native_dev(band_f64)

    def fence():
        while inflight:
            eng.infer_image_wait(inflight.pop(0))
        eng.synchronize()            # also surfaces UMX_ERR_RANGE of the split-precision path
        work.synchronize()
        if sharded and world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step):
        for _ in range(args.warmup):
            step()
        fence()
        eng.profile_enable(max(1, args.profile_every))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        fence()
        dt = time.perf_counter() - t0
        prof = eng.profile_read()
        eng.profile_enable(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof, res

    host_elapsed = host_prof = None
    host_sync = None
    if not args.resident_only:
        host_elapsed, host_prof, res_h = timed(step_host)
        if not sharded:
            # what ONE per-file driver call achieves: a synchronous umx_infer_image_raw with the drivers' intensity rescale, one
            # slide at a time
            sync_out = torch.empty((K, H, W), dtype=torch.uint8).pin_memory()

            def step_sync():
                eng._check(eng._L.umx_infer_image_raw(eng._ctx, host_u16.data_ptr(), 16, C_img, H, W, 1, float(mean), float(std),
                                                      umx.MODE_ACCUMULATE, sync_out.data_ptr()))

            def timed_sync(fn, n):
                fn()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                return time.perf_counter() - t0
            ns = max(3, min(args.steps, 10))
            dts = timed_sync(step_sync, ns)
            crc_sync = crc32_of(sync_out)
            # (round 3's form of the same call: the whole upload and a device reduction in front of the first tile)
            range_before = os.environ.get("UMX_HOST_RANGE")
            os.environ["UMX_HOST_RANGE"] = "0"
            nd = max(2, ns // 3)
            try:
                dtd = timed_sync(step_sync, nd)
            finally:
                if range_before is None:
                    del os.environ["UMX_HOST_RANGE"]
                else:
                    os.environ["UMX_HOST_RANGE"] = range_before
            host_sync = {"value": round(tiles_total * ns / dts, 2), "unit": "tiles/s", "ms_per_call": round(1e3 * dts / ns, 3),
                         "calls": ns, "device_range": {"value": round(tiles_total * nd / dtd, 2), "ms_per_call": round(1e3 * dtd / nd, 3)},
                         # (rescaled planes: another computation than `value`, so another CRC -- but the same one from both forms)
                         "crc32_u8_planes": crc_sync, "device_range_same_bytes": crc32_of(sync_out) == crc_sync,
                         "note": "synchronous umx_infer_image_raw(rescale=1), one slide at a time: the per-file driver's call -- the "
                                 "planes' (min, max) by host threads while the rows cross the bus, tiles start on the first slab; "
                                 "device_range (UMX_HOST_RANGE=0): whole upload + device min/max first, as in round 3 "
                                 "(`value` above streams two slides through submit / wait)"}
    res_elapsed, res_prof, res_r = timed(step_resident)
    # rounds 1-3 quoted the 2048-row band of the same slide (946 tiles: one rank's share at N = 8): kept beside the slide at N = 1
    weak_band = None
    if not sharded and args.workload == "wsi-synth256" and args.scaling == "strong" and host_elapsed is not None and H > 2048:
        Hb = 2048
        nb_r, nb_c, _, _ = eng.tile_grid(Hb, W)
        band_u16 = torch.empty((C_img, Hb, W), dtype=torch.int16).pin_memory()
        band_u16.copy_(host_u16[:, :Hb])
        band_outs = [torch.empty((K, Hb, W), dtype=torch.uint8).pin_memory() for _ in range(2)]

        def step_band():
            slot = step_band.n & 1
            step_band.n += 1
            if len(inflight) == 2:
                eng.infer_image_wait(inflight.pop(0))
            eng.infer_image_raw_submit(slot, band_u16.data_ptr(), 16, C_img, Hb, W, False, mean, std, band_outs[slot].data_ptr())
            inflight.append(slot)
        step_band.n = 0
        for _ in range(2):
            step_band()
        fence()
        nbs = 40
        t0 = time.perf_counter()
        for _ in range(nbs):
            step_band()
        fence()
        dtb = time.perf_counter() - t0
        weak_band = {"value": round(nb_r * nb_c * nbs / dtb, 2), "unit": "tiles/s", "tiles_per_step": int(nb_r * nb_c), "slide": [Hb, W],
                     "steps": nbs, "ms_per_step": round(1e3 * dtb / nbs, 3),
                     "note": "the first 2048 rows as a slide of their own, same host path (H2D + D2H inside): the workload of the "
                             "round-1..3 lines"}
    # the two paths must agree: uint8 planes of the host path == np.uint8(255 * fp16 planes) of the resident path
    with torch.cuda.stream(work):
        if not sharded:
            # the drivers' uint8 recipe (UnMicst1-5.py:848-854): np.uint8(255 * pm) with the product rounded to float16,
            # resize at the identity grid (u8 * (1/255) in float64), np.uint8(255 * .)
            want_u8 = None
            if host_elapsed is not None:
                first = (dev_out * 255.0).to(torch.uint8)
                want_u8 = (255.0 * (first.to(torch.float64) * (1.0 / 255))).to(torch.uint8).cpu()
            agree = None if want_u8 is None else bool(torch.equal(want_u8, host_outs[0]) and torch.equal(want_u8, host_outs[1]))
            # a checksum that depends on the data: CRC-32 of the uint8 planes the host path delivered, next to the same of the
            # drivers' uint8 recipe applied to the resident path's fp16 planes (equal iff every byte agrees)
            checksum = None if want_u8 is None else {"crc32_u8_planes_host_path": crc32_of(host_outs[0]),
                                                     "crc32_u8_planes_resident_path": crc32_of(want_u8)}
        else:
            agree = None
            if host_elapsed is not None:   # (both gathered stacks of the host path against the uint8 recipe on the resident path's)
                first = (res_r * 255.0).to(torch.uint8)
                want = (255.0 * (first.to(torch.float64) * (1.0 / 255))).to(torch.uint8)
                agree = bool(torch.equal(want, fulls_u8[0]) and torch.equal(want, fulls_u8[1])
                             and torch.equal(want[:, y0:y1].cpu(), host_owns[0]) and torch.equal(want[:, y0:y1].cpu(), host_owns[1]))
                del first, want
            checksum = None if host_elapsed is None else {"crc32_u8_own_rows_host_path": crc32_of(host_owns[0]),
                                                          "crc32_u8_stack_gathered": crc32_of(fulls_u8[0].cpu())}
    elapsed, prof = (host_elapsed, host_prof) if host_elapsed is not None else (res_elapsed, res_prof)
    ranks = ranks_block(dist if sharded else None, torch, hp, H, W, rank, agree)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = tiles_total * args.steps / elapsed
        roofline = roofline_of(prof, elapsed, eng, args.batch)
        if args.breakdown:
            print_breakdown(prof)
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(hp, blob, band_f64, mean, std, args.cpu_seconds)
        # two short legs the default line carries at N = 1: the exact-fp32 engine on the 2048-row band and the
        # training step of BASELINE configs[4]; `--no-legs` (or any non-default workload / N > 1) skips them
        f32 = None
        if legs and eng.precision != "f32":
            eng.close()
            del band_f64
            torch.cuda.empty_cache()
            f32 = f32_leg(torch, umx, hp, blob, local_rank, args.batch, host_u16, C_img, W, mean, std)
        configs = None
        if legs:
            if eng.precision == "f32":
                eng.close()
            del host_u16, host_outs, host_out, dev_out
            torch.cuda.empty_cache()
            configs = []
            for name, rows, nsteps in CONFIG_LEGS:
                try:
                    configs.append(config_leg(torch, umx, model, dev, local_rank, name, rows, nsteps, args.precision))
                except Exception as e:   # noqa: BLE001  (a leg must not take the headline line down with it: say what happened)
                    configs.append({"workload": name, "error": "%s: %s" % (type(e).__name__, e)})
        up_b = C_img * H * W * 2
        dn_b = K * H * W
        scope = ("H2D+D2H inside the timed region: uint16 planes up (%.0f MB), uint8 probability planes down (%.0f MB), "
                 "per launch group on two copy streams under the tile kernels (umx_infer_image_raw_submit / _wait, two slides "
                 "in flight; im2double on the device, no intensity rescale as in the solo driver)" % (up_b / 1e6, dn_b / 1e6)) if not sharded else (
                 "H2D+D2H inside the timed region: each rank uploads its uint16 band and downloads its own rows as uint8 planes "
                 "(umx_infer_image_sharded_raw_submit / umx_infer_image_wait, two slides in flight; im2double on the device, no intensity "
                 "rescale as in the solo driver); band halo exchange + slab-wise RCCL all-gather of the uint8 stack on every rank")
        if args.resident_only:
            scope = "slide resident in HBM, result left in HBM (kernel-only; NOT the section-8(d) metric)"
        line = {
            "metric": "tiles/sec (%dx%dx%d) whole-slide inference" % (hp.imSize, hp.imSize, hp.nChannels),
            "value": round(value, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16x3 (fp32 products as 3 binary16 MFMA products, fp32 accumulate)",
                      "f16f6": "f16x3, cross terms of the layers at <= 1/4 resolution as block-scaled fp6 (MX e2m3) MFMA products; fp32 accumulate"}[
                eng.precision], "data": "synthetic",
            "config": {"workload": "%s: %s hp (%s graph, seeded weights), %d-channel synthetic slide %dx%d, "
                                   "%d tiles/step, batch %d, fp16-compat stitch; %s" % (
                                       args.workload, key, "v2" if hp.graph else "legacy", C_img, H, W, tiles_total,
                                       args.batch, scope),
                       "tiles_per_step": tiles_total, "slide": [H, W], "flop_per_tile_as_written": hp.flops_per_tile(),
                       "flop_per_tile_executed_unpadded": umx.describe(hp)["flops_per_tile"], "checksum": checksum,
                       "host_path_equals_resident_path": agree},
            "resident": {"value": round(tiles_total * args.steps / res_elapsed, 2), "unit": "tiles/s",
                         "ms_per_step": round(1e3 * res_elapsed / args.steps, 3),
                         "note": "same slide already in HBM as float64, result left in HBM (no H2D / D2H): kernel-only"},
            "host_sync": host_sync, "weak_band": weak_band, "ranks": ranks, "shard_path": shard_path if sharded else None,
            "roofline": roofline, "cpu_baseline": cpu, "f32": f32, "train": train, "configs": configs,
            "workload_note": "the default workload is the full 16384 x 16384 slide at every N since round 4 (rounds 1-3 quoted its first "
                             "2048 rows: `weak_band` carries that line)",
        }
        print(json.dumps(line))
    eng.close()
    if sharded:
        dist.destroy_process_group()


def crc32_of(t):
    """CRC-32 of a host tensor's bytes (zlib; 0.4 s per GB): a checksum that depends on every output byte."""
    import zlib
    a = t.contiguous().view(-1).numpy()
    crc = 0
    step = 1 << 26
    for i in range(0, a.size, step):
        crc = zlib.crc32(a[i:i + step].tobytes() if a.dtype.itemsize != 1 else memoryview(a[i:i + step]), crc)
    return crc & 0xFFFFFFFF


CONFIG_LEGS = (   # (workload, rows, steps): the other BASELINE.json configs as short legs of the default N = 1 line
    ("solo-1024", 1024, 20),       # configs[1]: the reference's default tool (unmicstWrapper.py:55-57) on a 1024 x 1024 image
    ("duo-4096", 4096, 10),        # configs[2]
    ("solo-16384", 16384, 3),      # configs[3] as worded: the whole 16384 x 16384 slide, here on one GPU
    ("legacy-1024", 1024, 20),     # the only graph with weights in the reference's tree
)


def config_leg(torch, umx, model, dev, local_rank, name, H, steps, precision):
    """One BASELINE config through the same host path as the headline (uint16 planes up, uint8 planes down inside the timed region,
    two slides in flight): tiles/s, CRC-32 of the uint8 stack, the dominant kernel's fraction of the matrix peak -- or, for the
    16 - 64-channel legacy graph, which SURVEY section 2.2 classes HBM / LDS-bound, its compulsory bytes against 8 TB/s."""
    key, C_img, _, W = WORKLOADS[name]
    hp = model.KNOWN_HP[key]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = NORMALISATION[key]
    K = hp.nClasses
    batch = umx.auto_batch(hp)
    eng = umx.Engine(hp, blob, device=local_rank, max_batch=batch, precision=precision)
    try:
        npr, npc, _, _ = eng.tile_grid(H, W)
        band = torch.empty((C_img, H, W), dtype=torch.int16).pin_memory()
        rows_per = max(1, (1 << 26) // (W * C_img))          # (the slide is generated on the device in pieces of <= 64 M samples)
        for y in range(0, H, rows_per):
            n = min(rows_per, H - y)
            band[:, y:y + n].copy_(synth_rows_u16(torch, C_img, y, n, W, dev).to(torch.int16))
        outs = [torch.empty((K, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
        inflight = []

        def step(i):
            slot = i & 1
            if len(inflight) == 2:
                eng.infer_image_wait(inflight.pop(0))
            eng.infer_image_raw_submit(slot, band.data_ptr(), 16, C_img, H, W, False, mean, std, outs[slot].data_ptr())
            inflight.append(slot)

        def drain():
            while inflight:
                eng.infer_image_wait(inflight.pop(0))
            eng.synchronize()
        step(0)
        drain()
        eng.profile_enable(1)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        drain()
        dt = time.perf_counter() - t0
        prof = eng.profile_read()
        eng.profile_enable(False)
        tiles = npr * npc
        r = roofline_of(prof, dt, eng, batch)
        roof = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "share_of_step", "all_conv_launches") if k in r}
        if "as_matrix_work" in r:
            roof["as_matrix_work"] = r["as_matrix_work"]
        return {"workload": name, "model": key, "value": round(tiles * steps / dt, 2), "unit": "tiles/s", "tiles_per_step": int(tiles),
                "tile": [hp.imSize, hp.imSize, hp.nChannels], "slide": [int(H), int(W)], "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3),
                "batch": batch, "dtype": eng.precision, "crc32_u8_planes": crc32_of(outs[(steps - 1) & 1]), "roofline": roof}
    finally:
        eng.close()


def f32_leg(torch, umx, hp, blob, local_rank, batch, host_u16, C_img, W, mean, std, steps=4):
    """The exact-fp32 MFMA kernels (UMX_PREC_F32: v_mfma_f32_16x16x4_f32, bit for bit an fp32 fma chain) on the first 2048 rows
    of the same slide, same host path (H2D + D2H inside): what a model that leaves binary16's range falls back to."""
    Hb = min(2048, host_u16.shape[1])
    K = hp.nClasses
    eng = umx.Engine(hp, blob, device=local_rank, max_batch=batch, precision="f32")
    try:
        npr, npc, _, _ = eng.tile_grid(Hb, W)
        band = torch.empty((C_img, Hb, W), dtype=torch.int16).pin_memory()
        band.copy_(host_u16[:, :Hb])
        outs = [torch.empty((K, Hb, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
        inflight = []

        def step(i):
            slot = i & 1
            if len(inflight) == 2:
                eng.infer_image_wait(inflight.pop(0))
            eng.infer_image_raw_submit(slot, band.data_ptr(), 16, C_img, Hb, W, False, mean, std, outs[slot].data_ptr())
            inflight.append(slot)

        def drain():
            while inflight:
                eng.infer_image_wait(inflight.pop(0))
            eng.synchronize()
        step(0)
        drain()
        eng.profile_enable(1)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        drain()
        dt = time.perf_counter() - t0
        prof = eng.profile_read()
        eng.profile_enable(False)
        convs = [p for p in prof if p["kernel"].startswith("conv_")]
        ms = sum(p["total_ms"] for p in convs)
        fl = sum(p["flops"] for p in convs)
        tiles = npr * npc
        whole = hp.flops_per_tile() * tiles * steps / dt / 1e12
        return {"value": round(tiles * steps / dt, 2), "unit": "tiles/s", "tiles_per_step": int(tiles), "slide": [int(Hb), int(W)],
                "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3), "dtype": "f32",
                "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "all_conv_launches": {"achieved": round(fl / (ms * 1e-3) / 1e12, 2),
                                                   "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
                             "whole_step": {"achieved": round(whole, 2), "frac": round(whole / PEAK_F32_MFMA_TFLOPS, 4)}},
                "crc32_u8_planes": crc32_of(outs[0]),
                "note": "exact-fp32 MFMA engine (precision f32) on the first 2048 rows of the same slide, same host path"}
    finally:
        eng.close()


def train_leg(torch, dev, B=8, steps=100, warmup=10):
    """BASELINE.json configs[4] inside the default line: forward + backward + Adam on random 256x256x2 batches resident in HBM
    (the reference's sess.run([optOp ...]), UnMicst1-5.py:483-484), timed like `--workload train-synth256`."""
    from unmicst_amd import model, trainer
    hp = model.KNOWN_HP["synthetic-256"]
    blob = model.random_blob(hp, seed=20260101)
    tr = trainer.Trainer(hp, blob, trainer.duo_options(), batch=B, device=dev.index or 0)
    try:
        g = torch.Generator(device="cpu").manual_seed(20260101)
        nb = 4
        data = torch.randn((nb, B, hp.imSize, hp.imSize, hp.nChannels), generator=g).to(dev)
        cls = torch.randint(0, hp.nClasses, (nb, B, hp.imSize, hp.imSize), generator=g)
        labels = torch.nn.functional.one_hot(cls, hp.nClasses).float().to(dev)
        weights = (0.5 + 2.5 * torch.rand((nb, B, hp.imSize, hp.imSize, hp.nClasses), generator=g)).to(dev)
        torch.cuda.synchronize(dev)
        for i in range(warmup):
            tr.step_dev(data[i % nb].data_ptr(), labels[i % nb].data_ptr(), weights[i % nb].data_ptr())
        first = tr.loss()[0]
        t0 = time.perf_counter()
        for i in range(steps):
            j = (warmup + i) % nb
            tr.step_dev(data[j].data_ptr(), labels[j].data_ptr(), weights[j].data_ptr())
        last = tr.loss()[0]                        # synchronises the trainer's stream
        dt = time.perf_counter() - t0
        tflops = tr.flops_per_image * B * steps / dt / 1e12
        return {"value": round(B * steps / dt, 2), "unit": "images/s", "images_per_s": round(B * steps / dt, 2), "batch": B, "steps": steps,
                "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3), "flop_per_image": tr.flops_per_image,
                "loss_first": first, "loss_last": last,
                "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tflops / PEAK_F16_MFMA_TFLOPS, 4)},
                "dtype": "f16x3 convolutions, fp32 / fp64 elsewhere",
                "note": "forward + backward + Adam, synthetic-256 hp (duo regime), random 256x256x2 batches resident in HBM"}
    finally:
        tr.close()


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
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x3 convolutions (forward, input and weight gradients: fp32 products as 3 binary16 MFMA products, fp32 accumulate), "
                 "fp32 / fp64 elsewhere" if not os.environ.get("UMX_TRAIN_CONV_F32") else "f32 (weight gradients f16x3)",
        "data": "synthetic",
        "config": {"workload": "train-synth256: synthetic-256 hp (v2 graph, seeded weights), batch %d, duo regime "
                               "(Adam 6e-5, L2, dropout), random 256x256x2 batches resident in HBM" % B,
                   "batch": B, "flop_per_image": tr.flops_per_image, "loss_first": first, "loss_last": last,
                   "phase_ms_per_step": {k: round(v / max(ph["steps"], 1), 3) for k, v in ph.items() if k != "steps"}},
        # whole step against the fp32 matrix peak: forward + input-gradient + weight-gradient convolutions are
        # ~all of the algorithmic FLOPs; the element-wise BN/activation passes are HBM-bound and show up as lost fraction
        # whole step against the matrix peak of the dtype the convolutions run in (binary16 MFMA since round 4: three issued products
        # per algorithmic one, so 1/3 is the ceiling of `frac`); the fp32-MFMA fraction rounds 1-3 quoted rides along
        "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / PEAK_F16_MFMA_TFLOPS, 4), "traffic": None, "kernel": "whole training step",
                     "flop_per_launch": flops_step, "frac_of_fp32_mfma_peak": round(tflops / PEAK_F32_MFMA_TFLOPS, 4)},
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    tr.close()


def cpu_baseline_train(hp, blob, data, labels, weights, budget_s):
    """The training oracle (torch autograd on the host cores, float32) on the first images of the same batch, at the best of a
    sweep over thread counts (a GPU box shows every host core but grants a 16-core share: 128 threads ran 0.9 images/s)."""
    import torch
    from oracle import train_oracle as to
    o = to.duo_options()
    host = host_cpu_info()
    cap = int(host["affinity"] or host["logical"] or 1)
    quota = host["cgroup_quota_cores"]
    cands = {8, 16, 32, cap}
    if quota:
        cands |= {max(1, int(quota // 2)), max(1, int(round(quota))), max(1, int(2 * quota))}
    threads = sorted(t for t in cands if 1 <= t <= cap and (not quota or t <= 4 * quota))
    order = sorted(threads, key=lambda t: abs(t - (quota or min(cap, 32))))
    sweep, best, t_sweep = [], None, time.perf_counter()
    for nt in order:
        if best is not None and time.perf_counter() - t_sweep > 0.5 * budget_s:
            break
        torch.set_num_threads(nt)
        to.loss_and_grads(hp, blob, data[:1], labels[:1], weights[:1], o, 0, dtype=torch.float32)   # untimed: thread pool, caches
        t = time.perf_counter()
        to.loss_and_grads(hp, blob, data[:1], labels[:1], weights[:1], o, 0, dtype=torch.float32)
        rate = 1.0 / max(time.perf_counter() - t, 1e-6)
        sweep.append({"threads": nt, "images_per_s": round(rate, 3)})
        if best is None or rate > best[0]:
            best = (rate, nt)
    torch.set_num_threads(best[1])
    n = int(max(1, min(data.shape[0], 0.5 * budget_s * best[0])))
    t = time.perf_counter()
    to.loss_and_grads(hp, blob, data[:n], labels[:n], weights[:n], o, 0, dtype=torch.float32)
    dt = time.perf_counter() - t
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": best[1], "threads": best[1], "cores_granted": quota, "kind": "port",
            "sample": "forward+backward of %d image(s) of the same batch at the best of %d swept thread counts (oracle/train_oracle.py, "
                      "torch CPU float32, no optimiser update), %.1f s" % (n, len(sweep), dt),
            "host": host, "sweep": sweep}


def host_cpu_info():
    """What the host offers this process: logical CPUs, the affinity mask and the cgroup CPU quota (a GPU box shows every
    host core but may grant only a share of them)."""
    info = {"logical": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_quota_cores": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    info["cgroup_quota_cores"] = round(float(txt[0]) / float(txt[1]), 2)
            else:
                q = float(txt[0])
                if q > 0:
                    info["cgroup_quota_cores"] = round(q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    return info


def cpu_baseline(hp, blob, band_f64, mean, std, budget_s):
    """The reference's CPU path restated (TensorFlow is not installable here: SURVEY.md section 8c), timed on the host
    cores of the GPU box on a bounded sample of the same slide.  Two figures:
      value / best_effort   the UNet as torch-CPU float32 (oracle/train_oracle.py's restatement of the v2 graph; the C
                            oracle for the legacy graph), ONE pass that yields all classes, PI2D gather + normalise in numpy,
                            at the BEST of a sweep over thread count x batch size x memory format (every setting tried is
                            listed in `sweep`), on a sample of >= 64 tiles;
      reference_faithful    the reference's own loop shape: the Python PI2D tile loop with per-tile patchOutput, run once
                            PER CLASS (UnMicst1-5.py:697-707,845-848) on the same forward at the same settings -- a "tile"
                            still counts once, so this is ~nClasses x slower by construction."""
    import numpy as np
    import torch
    from oracle import oracle, pi2d_oracle
    P, m = hp.imSize, hp.margin
    sub = P - 2 * m
    hpB = max(1, int(getattr(hp, "batchSize", 0) or 8))
    host = host_cpu_info()
    cap = int(host["affinity"] or host["logical"] or 1)
    quota = host["cgroup_quota_cores"]
    cands = {8, 16, 32, 64, cap}
    if quota:
        cands |= {max(1, int(quota // 2)), max(1, int(round(quota))), max(1, int(2 * quota))}
    threads = sorted(t for t in cands if 1 <= t <= cap and (not quota or t <= 4 * quota))   # (far beyond the quota: minutes per batch)
    if hp.graph:
        from oracle import train_oracle as to
        T = to.split_blob(hp, np.asarray(blob, dtype=np.float64))
        Pm = {k: torch.tensor(v, dtype=torch.float32) for k, v in T.items()}
        opts = to.TrainOptions()

        def forward(x, fmt="channels_last"):
            with torch.no_grad():
                t = torch.from_numpy(np.ascontiguousarray(x))                       # NHWC: the oracle's NCHW view of it is channels_last
                if fmt == "contiguous":
                    t = t.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)      # ... or a contiguous NCHW tensor
                return to.forward(hp, Pm, t, opts, 0, training=False)[0].numpy()
        set_threads, what, formats = torch.set_num_threads, "torch CPU float32 (oracle/train_oracle.py)", ("channels_last", "contiguous")
    else:
        def forward(x, fmt=None):
            return oracle.forward(hp, blob, x)
        set_threads, what, formats = oracle.set_num_threads, "oracle/unet_oracle.c (OpenMP, double accumulate)", (None,)
    rows = min(band_f64.shape[1], 2 * sub + 2 * m)
    cols = min(band_f64.shape[2], 40 * sub + 2 * m)
    crop = band_f64[:, :rows, :cols].cpu().numpy()
    if hp.nChannels == 1:
        crop = crop[0]
    pi = pi2d_oracle.PI2DOracle(crop, P, m, "accumulate")

    def batch_of(t0, nb):
        return pi2d_oracle.normalised_batch(pi, t0, nb, hp.nChannels, mean, std, False)
    # ---- sweep (within half the budget): thread counts at the hp's batch size first, then batch size and memory format at the
    # best thread count; every setting = one untimed batch (shape-specific primitive caches) + one timed batch
    order = sorted(threads, key=lambda t: abs(t - (quota or min(cap, 32))))
    sweep, t_sweep = [], time.perf_counter()
    best = None

    def probe(nt, B, fmt):
        nonlocal best
        B = min(B, pi.num_patches)
        if any(e["threads"] == nt and e["batch"] == B and e["format"] == fmt for e in sweep):
            return
        if best is not None and time.perf_counter() - t_sweep > 0.5 * budget_s:
            return
        set_threads(nt)
        x = batch_of(0, B)
        forward(x, fmt)
        t = time.perf_counter()
        forward(x, fmt)
        rate = B / max(time.perf_counter() - t, 1e-6)
        sweep.append({"threads": nt, "batch": B, "format": fmt, "tiles_per_s": round(rate, 3)})
        if best is None or rate > best[0]:
            best = (rate, nt, B, fmt)
    for nt in order:
        probe(nt, hpB, formats[0])
    for B in (8, 32):
        for fmt in formats:
            probe(best[1], B, fmt)
    _, nt, B, fmt = best
    set_threads(nt)
    # ---- best effort at the best setting: >= 64 tiles (or every tile of the crop), one batched pass, all classes
    n = int(min(pi.num_patches, max(64, 0.25 * budget_s * best[0])))
    t = time.perf_counter()
    done = 0
    while done < n:
        nb = min(B, n - done)
        forward(batch_of(done, nb), fmt)
        done += nb
    dt = time.perf_counter() - t
    # ---- reference-faithful: the whole loop once per class on a crop of nf tiles
    per_tile = dt / n
    nf = int(max(1, min(pi.num_patches, 0.25 * budget_s / per_tile / hp.nClasses)))
    fr = max(1, min(2, nf // max(1, min(8, nf))))                # patch rows of the crop
    fc = max(1, min(nf // fr, 8))
    crop2 = crop[..., :fr * sub + 2 * m, :fc * sub + 2 * m]
    t = time.perf_counter()
    for k in range(hp.nClasses):
        pi2d_oracle.single_image_inference(crop2, lambda x: forward(x, fmt), P, hp.nChannels, mean, std, "accumulate", k, B)
    dtf = time.perf_counter() - t
    nft = pi2d_oracle.PI2DOracle(crop2, P, m, "accumulate").num_patches
    return {"value": round(n / dt, 3), "unit": "tiles/s", "cores": nt, "threads": nt, "cores_granted": quota, "kind": "port",
            "sample": "best effort: first %d tiles of the same slide at the best of %d swept settings (%d threads, batch %d, %s), one "
                      "pass for all classes (PI2D gather + normalise in numpy, UNet forward %s), %.1f s" % (
                          n, len(sweep), nt, B, fmt or "NHWC", what, dt),
            "host": host, "sweep": sweep,
            "reference_faithful": {"value": round(nft / dtf, 3), "unit": "tiles/s", "cores": nt, "threads": nt, "cores_granted": quota,
                                   "sample": "the reference's loop on a %d-tile crop: Python PI2D tile loop with float16 "
                                             "patchOutput, one full pass per class (%d passes), same forward and settings, %.1f s" % (
                                                 nft, hp.nClasses, dtf)}}


if __name__ == "__main__":
    main()
