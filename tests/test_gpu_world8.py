"""The target topology of SURVEY.md section 8(e) -- a world of EIGHT ranks -- executed on one GPU, and the failure behaviour of a
sharded world.

RCCL refuses two ranks on one device and the GPU boxes admit at most six processes on a card, so the eight ranks are eight libumx
contexts on eight host threads of ONE process (the library is re-entrant across contexts), their inter-rank operations routed through
umx_shard_init_transport: every send / recv / all-gather is staged through host memory behind the SAME function-pointer table
umx_shard_init fills with RCCL.  What runs is the product's band / halo / slab-gather / scatter schedule (umx_shard.hip) in the world
size the driver's 8-GPU run uses; what is checked is byte equality with the one-GPU entry points.  No reference counterpart
(UnMicst1-5.py:769: one device)."""
import ctypes
import os
import queue
import threading
import time

import numpy as np
import pytest

import helpers
from unmicst_amd import model, umx

pytestmark = pytest.mark.gpu


class ThreadWorld:
    """`world` ranks in one process: messages are host byte strings in per-(source, destination) queues, the all-gather meets at
    a barrier.  Every wait is bounded (`timeout` seconds): a rank whose peer never shows up gets an exception, which the ctypes
    guard of Engine.shard_init_transport turns into a non-zero status of the transport call."""

    def __init__(self, world, timeout=60.0):
        import torch
        self.world, self.timeout = world, timeout
        self.hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.parts = [None] * world
        self.calls = [dict(send=0, recv=0, all_gather=0) for _ in range(world)]
        self.fail_send = set()          # ranks whose next send raises (failure injection)

    def _sync(self, stream):
        assert self.hip.hipStreamSynchronize(ctypes.c_void_p(stream)) == 0

    def _d2h(self, ptr, n):
        b = ctypes.create_string_buffer(n)
        assert self.hip.hipMemcpy(b, ctypes.c_void_p(ptr), ctypes.c_size_t(n), 2) == 0
        return b

    def _h2d(self, ptr, b, n):
        assert self.hip.hipMemcpy(ctypes.c_void_p(ptr), b, ctypes.c_size_t(n), 1) == 0

    def transport(self, rank):
        pending = []

        def send(ptr, n, peer, stream):
            if rank in self.fail_send:
                self.fail_send.discard(rank)
                raise RuntimeError("injected: the link of rank %d is down" % rank)
            self._sync(stream)
            self.q[(rank, peer)].put((self._d2h(ptr, n), n))
            self.calls[rank]["send"] += 1

        def recv(ptr, n, peer, stream):
            self._sync(stream)
            pending.append((ptr, n, peer))
            self.calls[rank]["recv"] += 1

        def group_end():
            todo = list(pending)
            pending.clear()
            for ptr, n, peer in todo:
                b, m = self.q[(peer, rank)].get(timeout=self.timeout)
                assert m == n
                self._h2d(ptr, b, n)

        def all_gather(sp, rp, n, stream):
            self._sync(stream)
            self.parts[rank] = self._d2h(sp, n)
            self.barrier.wait()
            for r in range(self.world):
                self._h2d(rp + r * n, self.parts[r], n)
            self.barrier.wait()           # nobody overwrites its part before everyone has read it
            self.calls[rank]["all_gather"] += 1

        return dict(send=send, recv=recv, all_gather=all_gather, group_start=lambda: None, group_end=group_end)


def _run_ranks(world, body):
    """body(rank) on `world` threads; returns the per-rank results, re-raises the first exception."""
    out, errs = [None] * world, [None] * world

    def run(r):
        try:
            out[r] = body(r)
        except BaseException as e:   # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("nslabs", [1, 2])
def test_world_of_eight_on_one_gpu(nslabs):
    """19 patch rows over 8 ranks (bands of 3/3/3/2/2/2/2/2 rows -- uneven), both stitch element sizes of the device entry, the raw
    entry synchronous and with two slides in flight: every rank's gathered stack and own rows equal the one-GPU result byte for byte."""
    import torch
    world = 8
    hp = helpers.small_hps()["v2_duo_like"]                 # patch 32, margin 4, sub 24
    blob = model.random_blob(hp, seed=4)
    H, W = 24 * 19 - 7, 61
    K = hp.nClasses
    img = np.random.default_rng(8).random((2, H, W)) * 0.5
    raw = (np.random.default_rng(9).random((2, H, W)) * 50000).astype(np.uint16)
    raw2 = np.ascontiguousarray(raw[:, ::-1])
    rng = [(int(raw[c].min()), int(raw[c].max())) for c in range(2)]
    with umx.Engine(hp, blob, max_batch=8) as eng:
        want = {s: eng.infer_image(img, 0.2, 0.2, stitch=s) for s in (umx.STITCH_FP16_COMPAT, umx.STITCH_FP32)}
        want_raw = eng.infer_image_raw(raw, True, 0.2, 0.2, value_range=rng)
        want_raw2 = eng.infer_image_raw(raw2, True, 0.2, 0.2, value_range=rng)
        npr = eng.tile_grid(H, W)[0]
    assert npr == 19
    tw = ThreadWorld(world)

    def body(rank):
        torch.cuda.set_device(0)
        ok = []
        with umx.Engine(hp, blob, max_batch=8) as eng:
            eng.shard_init_transport(rank=rank, world=world, **tw.transport(rank))
            pl = eng.shard_plan(H, W, rank, world, nslabs)
            r0, r1, o0, o1 = pl["need_row0"], pl["need_row1"], pl["own_row0"], pl["own_row1"]
            for stitch, tdt, bits in ((umx.STITCH_FP16_COMPAT, torch.float16, np.uint16), (umx.STITCH_FP32, torch.float32, np.uint32)):
                band = torch.from_numpy(np.ascontiguousarray(img[:, r0:max(r1, r0 + 1)])).cuda()   # only this rank's rows
                full = torch.empty((K, H, W), dtype=tdt, device="cuda")
                torch.cuda.synchronize()
                eng.infer_image_sharded_dev(band.data_ptr(), 2, H, W, r0, band.shape[1], 0.2, 0.2, umx.MODE_ACCUMULATE, stitch, nslabs,
                                            full.data_ptr())
                eng.synchronize()
                torch.cuda.synchronize()
                ok.append(np.array_equal(full.cpu().numpy().view(bits), want[stitch].view(bits)))
            b1, b2 = np.ascontiguousarray(raw[:, r0:r1]), np.ascontiguousarray(raw2[:, r0:r1])
            full = torch.zeros((K, H, W), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            own = eng.infer_image_sharded_raw(b1, H, W, r0, rng, 0.2, 0.2, nslabs=nslabs, own_rows=o1 - o0, out_full_ptr=full.data_ptr())
            torch.cuda.synchronize()
            ok.append(np.array_equal(full.cpu().numpy(), want_raw) and np.array_equal(own, want_raw[:, o0:o1]))
            fulls = [torch.zeros((K, H, W), dtype=torch.uint8, device="cuda") for _ in range(2)]
            owns = [np.zeros((K, o1 - o0, W), np.uint8) for _ in range(2)]
            torch.cuda.synchronize()
            for slot, b in ((0, b1), (1, b2)):                                               # two slides in flight
                eng.infer_image_sharded_raw_submit(slot, b.ctypes.data, 16, 2, H, W, r0, b.shape[1], rng, 0.2, 0.2, umx.MODE_ACCUMULATE,
                                                   nslabs, owns[slot].ctypes.data, fulls[slot].data_ptr())
            eng.infer_image_wait(0)
            eng.infer_image_wait(1)
            torch.cuda.synchronize()
            ok.append(np.array_equal(fulls[0].cpu().numpy(), want_raw) and np.array_equal(fulls[1].cpu().numpy(), want_raw2)
                      and np.array_equal(owns[0], want_raw[:, o0:o1]) and np.array_equal(owns[1], want_raw2[:, o0:o1]))
        return (pl["patch_row0"], pl["patch_row1"]), ok

    res = _run_ranks(world, body)
    bands = [b for b, _ in res]
    assert [b - a for a, b in bands] == [3, 3, 3, 2, 2, 2, 2, 2] and bands[0][0] == 0 and bands[-1][1] == 19
    assert all(all(ok) for _, ok in res), res
    slides = 2 + 1 + 2
    for r in range(world):   # one halo row per slide and pair of neighbouring bands
        assert tw.calls[r]["send"] == (slides if r < world - 1 else 0) and tw.calls[r]["recv"] == (slides if r > 0 else 0)
        assert tw.calls[r]["all_gather"] >= slides


def test_a_failing_link_ends_both_ranks_with_a_status():
    """World of two; the transport's send of rank 0 fails in the middle of the first slide.  Rank 0 returns an error from the call that
    hit it; rank 1, whose recv never gets its halo row, returns one as soon as its (bounded) transport wait gives up -- no hang.  Both
    contexts then refuse further sharded calls until they are re-initialised, after which the world works again."""
    import torch
    world = 2
    hp = helpers.small_hps()["v2_duo_like"]
    blob = model.random_blob(hp, seed=4)
    H, W = 150, 61
    img = np.random.default_rng(3).random((2, H, W)) * 0.5
    with umx.Engine(hp, blob, max_batch=8) as eng:
        want = eng.infer_image(img, 0.2, 0.2)
    tw = ThreadWorld(world, timeout=4.0)
    tw.fail_send.add(0)
    tw2 = ThreadWorld(world)
    gate = threading.Barrier(world, timeout=60.0)

    def body(rank):
        torch.cuda.set_device(0)
        with umx.Engine(hp, blob, max_batch=8) as eng:
            eng.shard_init_transport(rank=rank, world=world, **tw.transport(rank))
            pl = eng.shard_plan(H, W, rank, world, 2)
            r0, r1 = pl["need_row0"], pl["need_row1"]
            band = torch.from_numpy(np.ascontiguousarray(img[:, r0:r1])).cuda()
            full = torch.empty((hp.nClasses, H, W), dtype=torch.float16, device="cuda")
            torch.cuda.synchronize()
            args = (band.data_ptr(), 2, H, W, r0, band.shape[1], 0.2, 0.2, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT, 2, full.data_ptr())
            t0 = time.time()
            with pytest.raises(umx.UmxError) as first:
                eng.infer_image_sharded_dev(*args)
            waited = time.time() - t0
            with pytest.raises(umx.UmxError) as again:          # the failed world is refused, not retried
                eng.infer_image_sharded_dev(*args)
            gate.wait()
            eng.shard_fini()
            eng.shard_init_transport(rank=rank, world=world, **tw2.transport(rank))
            eng.infer_image_sharded_dev(*args)
            eng.synchronize()
            torch.cuda.synchronize()
            same = np.array_equal(full.cpu().numpy().view(np.uint16), want.view(np.uint16))
        return waited, str(first.value), str(again.value), same

    t0 = time.time()
    res = _run_ranks(world, body)
    assert time.time() - t0 < 120
    for waited, first, again, same in res:
        assert waited < 30 and "failed earlier" in again and same, res
    assert "send" in res[0][1] and "recv" in res[1][1] or "group_end" in res[1][1], res


def test_sharded_entry_arguments_are_checked_before_anything_is_enqueued():
    """a band that does not cover the rows its patch rows read, a stitch code outside {fp16-compat, fp32}, std == 0, a wrong
    channel count and a NULL band are UMX_ERR_INVALID, not out-of-bounds device accesses."""
    import torch
    hp = helpers.small_hps()["v2_duo_like"]
    blob = model.random_blob(hp, seed=4)
    H, W = 100, 40
    tw = ThreadWorld(1)
    with umx.Engine(hp, blob, max_batch=8) as eng:
        eng.shard_init_transport(rank=0, world=1, **tw.transport(0))
        band = torch.zeros((2, H, W), dtype=torch.float64, device="cuda")
        full = torch.zeros((hp.nClasses, H, W), dtype=torch.float32, device="cuda")
        good = dict(band_ptr=band.data_ptr(), C=2, H=H, W=W, band_row0=0, band_rows=H, mean=0.2, std=0.2, mode=umx.MODE_ACCUMULATE,
                    stitch=umx.STITCH_FP16_COMPAT, nslabs=2, out_full_ptr=full.data_ptr())
        for bad in (dict(band_rows=H - 5), dict(band_row0=3, band_rows=H - 3), dict(stitch=3), dict(std=0.0), dict(C=3), dict(band_ptr=0),
                    dict(mode=7)):
            with pytest.raises(umx.UmxError) as e:
                eng.infer_image_sharded_dev(**dict(good, **bad))
            assert e.value.code == umx.ERR_INVALID, (bad, str(e.value))
        raw = np.zeros((2, H, W), np.uint16)
        own = np.zeros((hp.nClasses, H, W), np.uint8)
        with pytest.raises(umx.UmxError) as e:    # the raw entry: a band that is too short for its patch rows
            eng.infer_image_sharded_raw_submit(0, raw.ctypes.data, 16, 2, H, W, 0, H - 9, None, 0.2, 0.2, umx.MODE_ACCUMULATE, 2, own.ctypes.data, 0)
        assert e.value.code == umx.ERR_INVALID
        eng.infer_image_sharded_dev(**good)        # and the context is still usable
        eng.synchronize()
