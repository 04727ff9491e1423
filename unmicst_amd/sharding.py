"""Whole-slide inference sharded over the GPUs of one node: one process per GPU, torch.distributed (RCCL over xGMI).

The reference has no multi-device path at all (one process, one device: UnMicst1-5.py:769).  Tiles are independent
units; the only coupling is the overlap-add stitch (each output pixel is covered by <= 4 tiles), so:

* patch rows are split into contiguous bands, one per rank (``band_partition``);
* rank r computes the tile probabilities of its patch rows [pa, pb) from the image rows it holds;
* ONE exchange step: rank r sends the probabilities of its LAST patch row (computed first) to rank r+1 (the 2*margin
  image rows below a band boundary are covered by patch rows pb-1 and pb);
* rank r stitches the image rows it owns -- padded rows [pa*sub, pb*sub) -- visiting covering tiles in ascending
  global tile index, exactly like the single-GPU kernel, so the float16 result is bit-identical to a 1-GPU run;
* the stitched rows are all-gathered slab by slab (``dist.all_gather_into_tensor``, asynchronous, on equal-size padded
  slabs) while the next slab's tiles are being computed.

The engine object only needs ``hp``, ``tile_grid``, ``band_tiles_dev``, ``stitch_dev`` -- tests drive the same host
logic on CPU (gloo) with an oracle-backed stand-in.
"""
from __future__ import annotations

from typing import List, Tuple

from . import umx


def band_partition(npr: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced split of patch rows [0, npr) over ``world`` ranks (ranks >= npr get empty bands)."""
    active = min(world, npr)
    base, extra = divmod(npr, active) if active else (0, 0)
    out = []
    start = 0
    for r in range(world):
        n = (base + (1 if r < extra else 0)) if r < active else 0
        out.append((start, start + n))
        start += n
    return out


def owned_rows(pa: int, pb: int, npr: int, sub: int, margin: int, H: int) -> Tuple[int, int]:
    """Image rows stitched by the rank holding patch rows [pa, pb): padded rows [pa*sub, pb*sub) minus the margin,
    the first band starting at row 0 and the last one running to H."""
    if pa >= pb:
        return (0, 0)
    y0 = 0 if pa == 0 else min(H, max(0, pa * sub - margin))
    y1 = H if pb == npr else min(H, max(0, pb * sub - margin))
    return (y0, y1)


def needed_image_rows(pa: int, pb: int, sub: int, margin: int, patch: int, H: int) -> Tuple[int, int]:
    """Image rows that the tiles of patch rows [pa, pb) read (what a rank must hold in HBM)."""
    if pa >= pb:
        return (0, 0)
    return (max(0, pa * sub - margin), min(H, (pb - 1) * sub + patch - margin))


def _geometry(eng, H: int, W: int):
    hp = eng.hp
    npr, npc, _, _ = eng.tile_grid(H, W)
    margin = int(hp.imSize / 8)
    return npr, npc, margin, hp.imSize - 2 * margin


def infer_band_local(eng, d_image, mean: float, std: float, rank: int, world: int, mode: int, stitch: int):
    """What rank ``rank`` of ``world`` produces, computed stand-alone (the previous band's last patch row is
    recomputed instead of received).  Used by the 1-GPU bit-equality test and as the world_size == 1 path."""
    umx.require_torch_runtime("sharding.infer_band_local")
    import torch
    hp = eng.hp
    C, H, W = d_image.shape
    npr, npc, margin, sub = _geometry(eng, H, W)
    pa, pb = band_partition(npr, world)[rank]
    y0, y1 = owned_rows(pa, pb, npr, sub, margin, H)
    out_dtype = torch.float32 if stitch == umx.STITCH_FP32 else torch.float16
    out = torch.empty((hp.nClasses, y1 - y0, W), dtype=out_dtype, device=d_image.device)
    if pa >= pb or y1 <= y0:
        return out
    lo = max(0, pa - 1)
    probs = torch.empty(((pb - lo) * npc, hp.imSize, hp.imSize, hp.nClasses), dtype=torch.float32,
                        device=d_image.device)
    eng.band_tiles_dev(d_image.data_ptr(), C, H, W, 0, H, mean, std, lo, pb, probs.data_ptr())
    eng.stitch_dev(probs.data_ptr(), lo, pb, H, W, mode, stitch, y0, y1, out.data_ptr())
    eng.synchronize()
    return out


def _slab_cuts(pa: int, pb: int, n: int) -> List[int]:
    return [pa + ((pb - pa) * i) // n for i in range(n + 1)]


def slab_rows(pa: int, pb: int, npr: int, sub: int, margin: int, H: int, n: int, i: int) -> Tuple[int, int]:
    """Image rows of slab i of n of the band holding patch rows [pa, pb): the band's owned rows cut where patch-row
    cut c stops touching them -- image rows below c*sub - margin are covered by patch rows < c only."""
    if pa >= pb:
        return (0, 0)
    y0, y1 = owned_rows(pa, pb, npr, sub, margin, H)
    c = _slab_cuts(pa, pb, n)
    s0 = y0 if i == 0 else min(y1, max(y0, c[i] * sub - margin))
    s1 = y1 if i == n - 1 else min(y1, max(y0, c[i + 1] * sub - margin))
    return (s0, max(s0, s1))


class _Done:
    """A completed request (same surface as the Work objects torch.distributed returns)."""

    def __init__(self, fn=None):
        self._fn = fn

    def wait(self):
        if self._fn is not None:
            self._fn()
            self._fn = None


def _staged(t, group) -> bool:
    """GPU tensors over the gloo backend (the 2-ranks-on-one-GPU test: RCCL refuses two ranks on one device) travel
    through host memory; RCCL (backend "nccl") and CPU tensors over gloo go straight to torch.distributed."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _isend(t, dst, group):
    import torch.distributed as dist
    if _staged(t, group):
        h = t.cpu()
        w = dist.isend(h.view(-1), dst=dst, group=group)
        return _Done(lambda: (w.wait(), h))
    return dist.isend(t, dst=dst, group=group)


def _irecv(t, src, group):
    import torch
    import torch.distributed as dist
    if _staged(t, group):
        h = torch.empty(t.shape, dtype=t.dtype)
        w = dist.irecv(h.view(-1), src=src, group=group)
        return _Done(lambda: (w.wait(), t.copy_(h)))
    return dist.irecv(t, src=src, group=group)


def _all_gather(gathered, padded, group):
    import torch
    import torch.distributed as dist
    if _staged(padded, group):
        hp_, hg = padded.cpu(), torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(hg.view(torch.uint8), hp_.view(torch.uint8), group=group)
        gathered.copy_(hg)
        return _Done()
    if not padded.is_cuda:   # gloo (CPU tests): move the raw bytes, whatever the element type
        return dist.all_gather_into_tensor(gathered.view(torch.uint8), padded.view(torch.uint8), group=group, async_op=True)
    return dist.all_gather_into_tensor(gathered, padded, group=group, async_op=True)


def infer_image_sharded(eng, d_band, band_row0: int, H: int, W: int, mean: float, std: float, mode: int, stitch: int,
                        group=None, gather: bool = True, nslabs: int = 2, sync: bool = True):
    """Distributed whole-slide inference.  Every rank calls this with the image rows it holds.

    d_band: float64 tensor [C, band_rows, W] = image rows [band_row0, band_row0+band_rows) (must cover
    ``needed_image_rows`` of this rank's patch rows).  Returns the full [K, H, W] result on every rank
    (``gather=True``) or this rank's stitched band and its (y0, y1).

    ``sync=False`` leaves out the two host-side fences (``eng.synchronize()`` after the first patch row and at the end): the
    call then only ENQUEUES work -- a caller streaming several slides fences once, with ``eng.synchronize()``, which is also
    where an UMX_ERR_RANGE of the split-precision path surfaces.

    Stream contract (GPU): the caller runs this function under a dedicated ``torch.cuda.Stream`` and has handed that
    stream's handle to the engine (``eng.set_stream(stream.cuda_stream)``; the legacy default stream is refused there),
    so engine launches, torch copies and the hand-over to RCCL's stream are all ordered on ONE stream: a stitch is queued
    behind its slab's tiles, the copy into the gather buffer behind the stitch, ``work.wait()`` / ``req.wait()`` make that
    stream wait for the collective, and the caching allocator recycles a slab only behind the work that used it.
    Schedule -- communication rides under tile compute:
      1. the LAST patch row of the band is computed first and sent to the next rank (the one exchange step of the path);
      2. the band is then computed in ``nslabs`` slabs of patch rows; as soon as a slab's image rows are final they are
         stitched and their all-gather starts asynchronously, overlapping the next slab's tiles.
    The stitch still visits tiles in ascending global index, so the result is bit-identical to a single-GPU run.
    """
    umx.require_torch_runtime("sharding.infer_image_sharded")
    import torch
    import torch.distributed as dist
    hp = eng.hp
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    C, band_rows, _ = d_band.shape
    npr, npc, margin, sub = _geometry(eng, H, W)
    bands = band_partition(npr, world)
    pa, pb = bands[rank]
    active = [r for r in range(world) if bands[r][0] < bands[r][1]]
    y0, y1 = owned_rows(pa, pb, npr, sub, margin, H)
    dev = d_band.device
    out_dtype = torch.float32 if stitch == umx.STITCH_FP32 else torch.float16
    P, K = hp.imSize, hp.nClasses
    glob = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    # the same slab count on every rank (every rank issues the same sequence of collectives)
    n = max(1, min(int(nslabs), min(bands[r][1] - bands[r][0] for r in active)))

    has_prev = pa < pb and pa > 0
    has_next = pa < pb and pb < npr
    lo = pa - 1 if has_prev else pa
    probs = torch.empty((max(pb - lo, 0) * npc, P, P, K), dtype=torch.float32, device=dev)

    def tiles(r0, r1):   # patch rows [r0, r1) of this band -> probs
        if r1 > r0:
            eng.band_tiles_dev(d_band.data_ptr(), C, H, W, band_row0, band_rows, mean, std, r0, r1,
                               probs[(r0 - lo) * npc:].data_ptr())

    reqs = []
    if pa < pb:
        tiles(pb - 1, pb)                       # last patch row first: the next rank is waiting for it
        if sync:
            eng.synchronize()
    if has_next:
        reqs.append(_isend(probs[-npc:], glob(active[active.index(rank) + 1]), group))
    if has_prev:
        reqs.append(_irecv(probs[:npc], glob(active[active.index(rank) - 1]), group))

    cuts = _slab_cuts(pa, pb, n) if pa < pb else [0] * (n + 1)
    full = torch.empty((K, H, W), dtype=out_dtype, device=dev) if gather else None
    slabs, pending = [], []
    for i in range(n):
        tiles(cuts[i], min(cuts[i + 1], pb - 1))
        if i == 0:
            for r in reqs:                      # the previous rank's last patch row feeds the first rows of this band
                r.wait()
        s0, s1 = slab_rows(pa, pb, npr, sub, margin, H, n, i)
        slab = torch.empty((K, s1 - s0, W), dtype=out_dtype, device=dev)
        if s1 > s0:
            eng.stitch_dev(probs.data_ptr(), lo, pb, H, W, mode, stitch, s0, s1, slab.data_ptr())
        if not gather:
            slabs.append(slab)
            continue
        rows = [slab_rows(a, b, npr, sub, margin, H, n, i) for a, b in bands]
        mx = max(1, max(b - a for a, b in rows))
        padded = torch.zeros((K, mx, W), dtype=out_dtype, device=dev)
        padded[:, :s1 - s0] = slab
        gathered = torch.empty((world * K, mx, W), dtype=out_dtype, device=dev)   # concatenation along dim 0
        pending.append((_all_gather(gathered, padded, group), gathered, rows, mx))
    if not gather:
        eng.synchronize()
        return (torch.cat(slabs, dim=1) if slabs else torch.empty((K, 0, W), dtype=out_dtype, device=dev)), (y0, y1)
    for work, gathered, rows, mx in pending:
        work.wait()
        g4 = gathered.view(world, K, mx, W)
        for r, (a, b) in enumerate(rows):
            if b > a:
                full[:, a:b] = g4[r, :, :b - a]
    if sync:
        eng.synchronize()   # fence of the engine's stream: raises UmxError (e.g. UMX_ERR_RANGE of the split-precision path)
    return full
