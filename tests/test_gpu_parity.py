"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the committed golden fixtures.

Tolerances: per-tile softmax <= 1e-4 max-abs vs the fp32 CPU restatement (north_star);
fp16-compat stitch bit-exact vs the imported reference PI2D fed the same tile probabilities; end-to-end file
values <= 1 uint8 LSB vs the reference's bundled "UNet sample data" outputs.
"""
import numpy as np
import pytest

import helpers
from unmicst_amd import model, umx

pytestmark = pytest.mark.gpu

TILE_TOL = 1e-4  # north_star: probability maps within 1e-4 max-abs


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch


PRECS = ["f32", "f16x3"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", sorted(helpers.small_hps()))
def test_forward_tiles_matches_oracle(name, prec):
    from oracle import oracle
    hp = helpers.small_hps()[name]
    blob = model.random_blob(hp, seed=11)
    rng = np.random.default_rng(5)
    n = 5  # not a multiple of any tile-group size: exercises the image predicate
    x = rng.normal(size=(n, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
    ref = oracle.forward(hp, blob, x)
    with umx.Engine(hp, blob, max_batch=3, precision=prec) as eng:   # 5 tiles through batches of 3 + 2
        assert eng.precision == prec
        got = eng.forward_tiles(x)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= TILE_TOL, (name, err)
    assert np.allclose(got.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("prec", PRECS)
def test_forward_tiles_real_weights_nucleiDAPI(prec):
    """Legacy graph with the reference's shipped models/nucleiDAPI weights (5x5 kernels) on real sample tiles."""
    from oracle import oracle, pi2d_oracle
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw = helpers.load_sample_105()[0]
    I = helpers.legacy_preprocess(raw)
    pi = pi2d_oracle.PI2DOracle(I, hp.imSize, hp.margin, "accumulate")
    x = pi2d_oracle.normalised_batch(pi, 33, 6, 1, mean, std, False)
    ref = oracle.forward(hp, blob, x)
    with umx.Engine(hp, blob, max_batch=4, precision=prec) as eng:
        got = eng.forward_tiles(x)
    assert np.abs(got - ref).max() <= TILE_TOL


@pytest.mark.parametrize("name", helpers.PI2D_CASES)
def test_stitch_bit_exact_vs_reference_pi2d(name, torch_cuda):
    """umx_stitch_dev (fp16-compat) on the golden tile probabilities == outputs of the imported reference PI2D."""
    torch = torch_cuda
    c = helpers.load_pi2d_case(name)
    if c["margin"] != c["patch"] // 8:
        pytest.skip("engine ties margin to imSize/8 like singleImageInference does (UnMicst1-5.py:694)")
    img = c["image"]
    H, W = img.shape[-2:]
    hp = model.HParams(model.GRAPH_V2, c["patch"], 1, c["nclass"], 4, 1, 3, 0)
    with umx.Engine(hp, model.random_blob(hp), max_batch=2) as eng:
        npr, npc, nrpi, ncpi = eng.tile_grid(H, W)
        assert (nrpi, ncpi) == (c["nrpi"], c["ncpi"]) and npr * npc == c["probs"].shape[0]
        probs = torch.from_numpy(c["probs"]).cuda()
        out = torch.empty((c["nclass"], H, W), dtype=torch.float16, device="cuda")
        mode = umx.MODE_REPLACE if c["mode"] == "replace" else umx.MODE_ACCUMULATE
        eng.stitch_dev(probs.data_ptr(), 0, npr, H, W, mode, umx.STITCH_FP16_COMPAT, 0, H, out.data_ptr())
        eng.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint16), c["stitched"].view(np.uint16)), name
        # banded stitch (what a rank of the sharded path does) must agree bit for bit with the full one
        if npr >= 2:
            sub = c["patch"] - 2 * c["margin"]
            y_split = min(H, 1 * sub - c["margin"])   # rows owned by patch row 0 when it is a band of its own
            if 0 < y_split < H:
                top = torch.empty((c["nclass"], y_split, W), dtype=torch.float16, device="cuda")
                bot = torch.empty((c["nclass"], H - y_split, W), dtype=torch.float16, device="cuda")
                eng.stitch_dev(probs.data_ptr(), 0, 1, H, W, mode, 0, 0, y_split, top.data_ptr())
                eng.stitch_dev(probs.data_ptr(), 0, npr, H, W, mode, 0, y_split, H, bot.data_ptr())
                eng.synchronize()
                both = np.concatenate([top.cpu().numpy(), bot.cpu().numpy()], axis=1)
                assert np.array_equal(both.view(np.uint16), c["stitched"].view(np.uint16))


@pytest.mark.parametrize("prec", PRECS)
def test_end_to_end_reference_sample_data(prec):
    """umx_infer_image with models/nucleiDAPI on 'UNet sample data' 105.tif vs the reference's bundled outputs."""
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw, g_cont, g_raw, g_nuc = helpers.load_sample_105()
    I = helpers.legacy_preprocess(raw)
    with umx.Engine(hp, blob, max_batch=32, precision=prec) as eng:
        planes = eng.infer_image(I, mean, std)
    assert planes.dtype == np.float16 and planes.shape == (3,) + I.shape
    for k, gold in ((1, g_cont), (2, g_nuc)):
        pm = np.uint8(255 * planes[k].astype(np.float64))
        d = np.abs(pm.astype(int) - gold.astype(int))
        assert d.max() <= 1, (k, d.max())
        assert (d == 0).mean() > 0.98


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name,shape", [("v2_duo_like", (2, 70, 45)), ("v2_solo_like", (53, 90)), ("legacy_k5", (40, 33))])
def test_infer_image_matches_oracle_loop(name, shape, prec):
    """Whole-image path (gather+normalise, UNet, stitch) vs the reference-equivalent oracle loop, per class."""
    from oracle import oracle
    hp = helpers.small_hps()[name]
    blob = model.random_blob(hp, seed=3)
    img = np.random.default_rng(9).random(shape) * 0.6
    mean, std = 0.21, 0.17
    with umx.Engine(hp, blob, max_batch=4, precision=prec) as eng:
        got16 = eng.infer_image(img, mean, std)
        got32 = eng.infer_image(img, mean, std, stitch=umx.STITCH_FP32)
        rep = eng.infer_image(img, mean, std, mode=umx.MODE_REPLACE)
    for k in range(hp.nClasses):
        ref = oracle.single_image_inference(hp, blob, img, mean, std, "accumulate", k, batch_size=8)
        assert ref.dtype == np.float16
        # tile probabilities agree to ~1e-6, so the fp16 planes may differ by one fp16 ulp (4.9e-4 below 1.0)
        assert np.abs(got16[k].astype(np.float32) - ref.astype(np.float32)).max() <= 1e-3
        assert (got16[k].view(np.uint16) == ref.view(np.uint16)).mean() > 0.98
        assert np.abs(got32[k] - ref.astype(np.float32)).max() <= 1e-3
        refr = oracle.single_image_inference(hp, blob, img, mean, std, "replace", k, batch_size=8)
        assert np.abs(rep[k].astype(np.float32) - refr.astype(np.float32)).max() <= 1e-3


def test_band_sharding_is_bit_equal_to_single_pass(torch_cuda):
    """Sharded whole-slide path on ONE GPU: two bands of patch rows + halo tiles == single pass, bit for bit."""
    torch = torch_cuda
    from unmicst_amd import sharding
    hp = helpers.small_hps()["v2_duo_like"]
    blob = model.random_blob(hp, seed=4)
    img = np.random.default_rng(2).random((2, 150, 61)) * 0.5
    with umx.Engine(hp, blob, max_batch=8) as eng:
        full = eng.infer_image(img, 0.2, 0.2)
        d_img = torch.from_numpy(img).cuda()
        parts = []
        for rank in range(3):
            parts.append(sharding.infer_band_local(eng, d_img, 0.2, 0.2, rank, 3, umx.MODE_ACCUMULATE,
                                                   umx.STITCH_FP16_COMPAT))
        got = np.concatenate([p.cpu().numpy() for p in parts], axis=1)
    assert np.array_equal(got.view(np.uint16), full.view(np.uint16))


def test_errors_are_reported_not_raised_in_c():
    hp = helpers.small_hps()["v2_solo_like"]
    blob = model.random_blob(hp)
    with pytest.raises(umx.UmxError) as e:
        umx.Engine(hp, blob[:-1])
    assert e.value.code == 2  # UMX_ERR_BLOB, the analogue of tf's NotFoundError on restore
    with umx.Engine(hp, blob, max_batch=2) as eng:
        with pytest.raises(umx.UmxError):
            eng.infer_image(np.zeros((3, 20, 20)), 0.1, 0.1)  # 3 channels into a 1-channel model
        with pytest.raises(umx.UmxError):
            eng.infer_image(np.zeros((20, 20)), 0.1, 0.0)     # std == 0


def test_sharded_entry_point_with_rccl_world_of_one(tmp_path):
    """bench.py's N>1 code path (sharding.infer_image_sharded: band tiles, halo exchange, RCCL all-gather) driven with the
    real engine and the nccl backend in a world of one rank; the result must equal umx_infer_image bit for bit.  (Worlds of
    2 and 3 ranks run on CPU over gloo in tests/test_sharding_cpu.py; 8 GPUs are the driver's to launch.)"""
    import subprocess
    import sys
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import helpers
from unmicst_amd import model, sharding, umx
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hp = helpers.small_hps()["v2_duo_like"]
blob = model.random_blob(hp, seed=4)
img = np.random.default_rng(2).random((2, 150, 61)) * 0.5
with umx.Engine(hp, blob, max_batch=8) as eng:
    want = eng.infer_image(img, 0.2, 0.2)
    work = torch.cuda.Stream()
    eng.set_stream(work.cuda_stream)
    with torch.cuda.stream(work):
        band = torch.from_numpy(img).cuda()
        got = sharding.infer_image_sharded(eng, band, 0, 150, 61, 0.2, 0.2, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    try:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)   # the legacy default stream (handle 0) is refused
        raise SystemExit("set_stream(0) was accepted")
    except ValueError:
        pass
dist.destroy_process_group()
print("sharded ok")
''' % (helpers.ROOT, helpers.ROOT)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stderr[-3000:]


def test_range_overflow_is_reported_on_every_path_and_the_facade_falls_back(torch_cuda):
    """UMX_ERR_RANGE (an activation beyond binary16's range in the split-precision kernels) must surface from the host
    entry point, from the device entry points' fence (umx_synchronize) and from the sharded path -- and the UNet2D facade
    must then rebuild the engine with the exact-fp32 kernels and succeed, as the reference's fp32 graph would."""
    import torch
    from unmicst_amd import sharding
    from unmicst_amd.unet2d import UNet2D
    hp = helpers.small_hps()["v2_duo_like"]
    blob = model.random_blob(hp, seed=4)
    img = np.random.default_rng(2).random((2, 70, 45)) * 0.5 + 0.25
    tiny_std = 1e-7                      # (v - mean) / std ~ 5e6: far outside binary16
    with umx.Engine(hp, blob, max_batch=8, precision="f16x3") as eng:
        with pytest.raises(umx.UmxError) as e1:
            eng.infer_image(img, 0.0, tiny_std)
        assert e1.value.code == umx.ERR_RANGE
        good = eng.infer_image(img, 0.2, 0.2)                       # the context stays usable, the flag is cleared
        assert np.isfinite(good.astype(np.float32)).all()
        work = torch.cuda.Stream()
        eng.set_stream(work.cuda_stream)
        with torch.cuda.stream(work):
            d = torch.from_numpy(img).cuda()
            out = torch.empty((hp.nClasses, 70, 45), dtype=torch.float16, device="cuda")
            eng.infer_image_dev(d.data_ptr(), 2, 70, 45, 0.0, tiny_std, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT,
                                out.data_ptr())
            with pytest.raises(umx.UmxError) as e2:
                eng.synchronize()
            assert e2.value.code == umx.ERR_RANGE
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import helpers
from unmicst_amd import model, sharding, umx
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29587")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hp = helpers.small_hps()["v2_duo_like"]
blob = model.random_blob(hp, seed=4)
img = np.random.default_rng(2).random((2, 150, 61)) * 0.5 + 0.25
with umx.Engine(hp, blob, max_batch=8, precision="f16x3") as eng:
    work = torch.cuda.Stream()
    eng.set_stream(work.cuda_stream)
    with torch.cuda.stream(work):
        band = torch.from_numpy(img).cuda()
        try:
            sharding.infer_image_sharded(eng, band, 0, 150, 61, 0.0, 1e-7, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT)
            print("no error")
        except umx.UmxError as e:
            print("sharded raised code %%d" %% e.code)
dist.destroy_process_group()
''' % (helpers.ROOT, helpers.ROOT)
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert "sharded raised code 6" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    # the facade: same overflow, transparent switch to the fp32 kernels
    UNet2D.setupWithArtefacts(model.ModelArtefacts(hp, blob, 0.0, tiny_std))
    try:
        assert UNet2D.Engine.precision == "f16x3"
        plane = UNet2D.singleImageInference(img, "accumulate", 1)
        assert UNet2D.Engine.precision == "f32" and plane.shape == (70, 45)
        assert np.isfinite(plane.astype(np.float32)).all()
    finally:
        UNet2D.singleImageInferenceCleanup()


def test_native_sharded_entry_point_world_of_one():
    """umx_infer_image_sharded_dev (RCCL inside libumx: communicator from umx_shard_unique_id / umx_shard_init, band tiles,
    slab-wise ncclAllGather on the library's communication stream) in a world of one rank, 1 and 3 slabs, fp16-compat and
    fp32 stitch: bit-equal to umx_infer_image; then umx_infer_image_sharded_raw[_submit] over the same communicator.  (The band
    geometry is checked against the torch.distributed schedule on CPU, tests/test_sharding_cpu.py; RCCL refuses two ranks on one
    device, so worlds > 1 are the driver's 8-GPU run.)"""
    import subprocess
    import sys
    script = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import helpers
from unmicst_amd import model, umx
hp = helpers.small_hps()["v2_duo_like"]
blob = model.random_blob(hp, seed=4)
H, W = 233, 97
img = np.random.default_rng(2).random((2, H, W)) * 0.5
with umx.Engine(hp, blob, max_batch=8) as eng:
    eng.shard_init(umx.Engine.shard_unique_id(), 0, 1)
    band = torch.from_numpy(img).cuda()
    for stitch, dt in ((umx.STITCH_FP16_COMPAT, torch.float16), (umx.STITCH_FP32, torch.float32)):
        want = eng.infer_image(img, 0.2, 0.2, stitch=stitch)
        for nslabs in (1, 3):
            out = torch.zeros((hp.nClasses, H, W), dtype=dt, device="cuda")
            eng.infer_image_sharded_dev(band.data_ptr(), 2, H, W, 0, H, 0.2, 0.2, umx.MODE_ACCUMULATE, stitch, nslabs,
                                        out.data_ptr())
            eng.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint8), want.view(np.uint8)), (stitch, nslabs)
    # the raw entry over the same RCCL communicator (ncclAllGather of the uint8 slabs in a world of one), synchronous and two slides in
    # flight, with and without the drivers' rescale: byte for byte umx_infer_image_raw[_range]
    K = hp.nClasses
    raw = (np.random.default_rng(3).random((2, H, W)) * 60000).astype(np.uint16)
    for rescale in (False, True):
        rng = [(int(raw[c].min()), int(raw[c].max())) for c in range(2)] if rescale else None
        want = eng.infer_image_raw(raw, rescale, 0.2, 0.2, value_range=rng)
        for nslabs in (1, 3):
            full = torch.zeros((K, H, W), dtype=torch.uint8, device="cuda")
            own = eng.infer_image_sharded_raw(raw, H, W, 0, rng, 0.2, 0.2, nslabs=nslabs, own_rows=H, out_full_ptr=full.data_ptr())
            assert np.array_equal(own, want) and np.array_equal(full.cpu().numpy(), want), (rescale, nslabs)
        owns = [np.zeros((K, H, W), np.uint8) for _ in range(2)]
        for slot in (0, 1):
            eng.infer_image_sharded_raw_submit(slot, raw.ctypes.data, 16, 2, H, W, 0, H, rng, 0.2, 0.2, umx.MODE_ACCUMULATE, 2,
                                               owns[slot].ctypes.data, 0)          # (gathered stack into the library's own buffer)
        eng.infer_image_wait(0); eng.infer_image_wait(1)
        assert np.array_equal(owns[0], want) and np.array_equal(owns[1], want), rescale
    try:                                                # the device entry refuses to run under a submitted call
        eng.infer_image_sharded_raw_submit(0, raw.ctypes.data, 16, 2, H, W, 0, H, None, 0.2, 0.2, umx.MODE_ACCUMULATE, 2, owns[0].ctypes.data, 0)
        refused = False
        try:
            eng.infer_image_sharded_dev(band.data_ptr(), 2, H, W, 0, H, 0.2, 0.2, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT, 1, out.data_ptr())
        except umx.UmxError:
            refused = True
        assert refused
    finally:
        eng.infer_image_wait(0)
print("native sharded ok")
''' % (helpers.ROOT, helpers.ROOT)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "native sharded ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_two_and_three_ranks_share_one_gpu(world, tmp_path):
    """The N>1 path with the REAL engine: `world` processes, all on cuda:0 (RCCL refuses two ranks on one device, so the
    messages travel over gloo through host memory -- sharding._staged), each holding only its band of the image on the
    device; every rank's gathered result must equal the single-process umx_infer_image bit for bit."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import helpers
from unmicst_amd import model, sharding, umx
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
hp = helpers.small_hps()["v2_duo_like"]
blob = model.random_blob(hp, seed=4)
H, W = 233, 97
img = np.random.default_rng(2).random((2, H, W)) * 0.5
with umx.Engine(hp, blob, max_batch=8) as eng:
    want = eng.infer_image(img, 0.2, 0.2)
    npr, npc, _, _ = eng.tile_grid(H, W)
    m = hp.margin; sub = hp.imSize - 2 * m
    pa, pb = sharding.band_partition(npr, world)[rank]
    r0, r1 = sharding.needed_image_rows(pa, pb, sub, m, hp.imSize, H)
    work = torch.cuda.Stream()
    eng.set_stream(work.cuda_stream)
    with torch.cuda.stream(work):
        band = torch.from_numpy(np.ascontiguousarray(img[:, r0:max(r1, r0 + 1)])).cuda()   # only this rank's rows
        got = sharding.infer_image_sharded(eng, band, r0, H, W, 0.2, 0.2, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT,
                                           nslabs=2)
    torch.cuda.synchronize()
    same = np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    print("rank %%d of %%d: bands %%s equal=%%s" %% (rank, world, (pa, pb), same), flush=True)
    assert same
dist.destroy_process_group()
''' % (helpers.ROOT, helpers.ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29590 + world), str(script)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("equal=True") == world, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("world,nslabs", [(2, 2), (3, 1), (3, 3)])
def test_native_sharded_entry_point_in_worlds_of_two_and_three(world, nslabs, tmp_path):
    """umx_infer_image_sharded_dev -- the in-library band / halo / slab-gather / scatter schedule -- with its inter-rank operations
    routed through umx_shard_init_transport: `world` processes on cuda:0, every send / recv / all-gather staged through host memory
    over gloo behind the SAME function-pointer table umx_shard_init fills with RCCL (which refuses two ranks on one device).  Each
    rank holds only its band; every rank's full result must equal the single-process umx_infer_image bit for bit, in both stitch
    modes' element sizes (fp16-compat here, fp32 below), for two slides in a row (buffer reuse).  Then the raw entry
    (umx_infer_image_sharded_raw[_submit]) against umx_infer_image_raw[_range], byte for byte, also with two slides in flight."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(r'''
import ctypes, os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import helpers
from unmicst_amd import model, sharding, umx
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nslabs = int(sys.argv[1])
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
vp, sz = ctypes.c_void_p, ctypes.c_size_t
def sync(stream): assert hip.hipStreamSynchronize(vp(stream)) == 0
def d2h(ptr, n):
    b = torch.empty(n, dtype=torch.uint8)
    assert hip.hipMemcpy(vp(b.data_ptr()), vp(ptr), sz(n), 2) == 0
    return b
def h2d(ptr, b): assert hip.hipMemcpy(vp(ptr), vp(b.data_ptr()), sz(b.numel()), 1) == 0
ops, calls = [], {"send": 0, "recv": 0, "all_gather": 0}
def send(ptr, n, peer, stream):
    sync(stream); b = d2h(ptr, n); ops.append((dist.isend(b, peer), None, b)); calls["send"] += 1
def recv(ptr, n, peer, stream):
    sync(stream); b = torch.empty(n, dtype=torch.uint8); ops.append((dist.irecv(b, peer), ptr, b)); calls["recv"] += 1
def group_end():
    for w, ptr, b in ops:
        w.wait()
        if ptr is not None: h2d(ptr, b)
    ops.clear()
def all_gather(sp, rp, n, stream):
    sync(stream); b = d2h(sp, n); parts = [torch.empty(n, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(parts, b); h2d(rp, torch.cat(parts)); calls["all_gather"] += 1
hp = helpers.small_hps()["v2_duo_like"]
blob = model.random_blob(hp, seed=4)
ok = True
with umx.Engine(hp, blob, max_batch=8) as eng:
    eng.shard_init_transport(send, recv, all_gather, rank, world, group_start=lambda: None, group_end=group_end)
    for (H, W, stitch, dt) in ((233, 97, umx.STITCH_FP16_COMPAT, np.float16), (150, 120, umx.STITCH_FP32, np.float32)):
        img = np.random.default_rng(H).random((2, H, W)) * 0.5
        want = eng.infer_image(img, 0.2, 0.2, stitch=stitch)
        pl = eng.shard_plan(H, W, rank, world, nslabs)
        r0, r1 = pl["need_row0"], pl["need_row1"]
        band = torch.from_numpy(np.ascontiguousarray(img[:, r0:max(r1, r0 + 1)])).cuda()     # only this rank's rows
        full = torch.empty((hp.nClasses, H, W), dtype=torch.float16 if dt == np.float16 else torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.infer_image_sharded_dev(band.data_ptr(), 2, H, W, r0, band.shape[1], 0.2, 0.2, umx.MODE_ACCUMULATE, stitch, nslabs,
                                    full.data_ptr())
        eng.synchronize()
        torch.cuda.synchronize()
        got = full.cpu().numpy()
        same = np.array_equal(got.view(np.uint16 if dt == np.float16 else np.uint32), want.view(np.uint16 if dt == np.float16 else np.uint32))
        ok = ok and same
        print("rank %%d of %%d: %%d x %%d patch rows [%%d, %%d) equal=%%s calls=%%s" %% (rank, world, H, W, pl["patch_row0"], pl["patch_row1"], same, calls), flush=True)
    # the raw entry (umx_infer_image_sharded_raw: raw uint16 / uint8 band up from the host, im2double [+ the drivers' rescale to the WHOLE
    # planes' range] in the tile gather, slabs cast to uint8 before the gather, own rows down to the host) against the one-GPU
    # umx_infer_image_raw[_range] of the whole slide: the gathered stack on every rank and the rank's own rows, byte for byte --
    # synchronous, then two slides in flight on the two slots (per-slot gather buffers)
    K = hp.nClasses
    # (the last slide has two patch rows: in a world of three the third rank's band is empty -- it still takes part in every collective)
    for (H, W, dt, rescale) in ((233, 97, np.uint16, False), (150, 120, np.uint16, True), (97, 61, np.uint8, True), (40, 75, np.uint16, False)):
        top = 40000 if dt == np.uint16 else 200
        raw = (np.random.default_rng(H + 1).random((2, H, W)) * top).astype(dt)
        rng = [(int(raw[c].min()), int(raw[c].max())) for c in range(2)] if rescale else None
        want = eng.infer_image_raw(raw, rescale, 0.2, 0.2, value_range=rng)
        pl = eng.shard_plan(H, W, rank, world, nslabs)
        r0, r1, o0, o1 = pl["need_row0"], pl["need_row1"], pl["own_row0"], pl["own_row1"]
        band = np.ascontiguousarray(raw[:, r0:r1])                                           # only this rank's rows
        full = torch.zeros((K, H, W), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        own = eng.infer_image_sharded_raw(band, H, W, r0, rng, 0.2, 0.2, nslabs=nslabs, own_rows=o1 - o0, out_full_ptr=full.data_ptr())
        torch.cuda.synchronize()
        same = np.array_equal(full.cpu().numpy(), want) and np.array_equal(own, want[:, o0:o1])
        # two slides in flight: slot 1 is submitted while slot 0 is still running; the second slide is the first one flipped
        raw2 = np.ascontiguousarray(raw[:, ::-1])
        rng2 = rng
        want2 = eng.infer_image_raw(raw2, rescale, 0.2, 0.2, value_range=rng2)
        band2 = np.ascontiguousarray(raw2[:, r0:r1])
        fulls = [torch.zeros((K, H, W), dtype=torch.uint8, device="cuda") for _ in range(2)]
        owns = [np.zeros((K, o1 - o0, W), np.uint8) for _ in range(2)]
        torch.cuda.synchronize()
        for slot, b in ((0, band), (1, band2)):
            eng.infer_image_sharded_raw_submit(slot, b.ctypes.data if b.size else 0, 8 * b.dtype.itemsize, 2, H, W, r0, b.shape[1], rng, 0.2, 0.2,
                                               umx.MODE_ACCUMULATE, nslabs, owns[slot].ctypes.data if owns[slot].size else 0, fulls[slot].data_ptr())
        eng.infer_image_wait(0); eng.infer_image_wait(1)
        torch.cuda.synchronize()
        same2 = (np.array_equal(fulls[0].cpu().numpy(), want) and np.array_equal(fulls[1].cpu().numpy(), want2)
                 and np.array_equal(owns[0], want[:, o0:o1]) and np.array_equal(owns[1], want2[:, o0:o1]))
        ok = ok and same and same2
        print("rank %%d of %%d: raw %%s %%d x %%d rescale=%%s own rows [%%d, %%d) equal=%%s in-flight equal=%%s" %% (rank, world, dt.__name__, H, W, rescale, o0, o1, same, same2), flush=True)
# an engine whose tile gather reads float64 (exact-fp32 precision): the raw entry uploads the band whole and converts it first
with umx.Engine(hp, blob, max_batch=8, precision="f32") as eng:
    eng.shard_init_transport(send, recv, all_gather, rank, world, group_start=lambda: None, group_end=group_end)
    H, W = 120, 88
    raw = (np.random.default_rng(5).random((2, H, W)) * 50000).astype(np.uint16)
    rng = [(int(raw[c].min()), int(raw[c].max())) for c in range(2)]
    want = eng.infer_image_raw(raw, True, 0.2, 0.2, value_range=rng)
    pl = eng.shard_plan(H, W, rank, world, nslabs)
    r0, r1, o0, o1 = pl["need_row0"], pl["need_row1"], pl["own_row0"], pl["own_row1"]
    full = torch.zeros((hp.nClasses, H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    own = eng.infer_image_sharded_raw(np.ascontiguousarray(raw[:, r0:r1]), H, W, r0, rng, 0.2, 0.2, nslabs=nslabs, own_rows=o1 - o0,
                                      out_full_ptr=full.data_ptr())
    torch.cuda.synchronize()
    same = np.array_equal(full.cpu().numpy(), want) and np.array_equal(own, want[:, o0:o1])
    ok = ok and same
    print("rank %%d of %%d: raw entry on the f32 engine equal=%%s" %% (rank, world, same), flush=True)
slides = 2 + 3 * 3 + 1
# one halo row per slide between neighbouring NON-EMPTY bands; the three 40-row slides have two patch rows: rank 0 -> rank 1 only
assert calls["send"] == (slides if rank < world - 1 else 0) + (3 if rank == 0 else 0), calls
assert calls["recv"] == (slides if rank > 0 else 0) + (3 if rank == 1 else 0), calls
assert calls["all_gather"] >= slides + 3
assert ok
print("rank %%d native transport ok" %% rank, flush=True)
dist.barrier()
dist.destroy_process_group()
''' % (helpers.ROOT, helpers.ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29600 + 3 * world + nslabs), str(script), str(nslabs)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("native transport ok") == world, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("key", ["nucleiDAPI1-5", "nucleiDAPILAMIN"])
def test_forward_tiles_shipped_hyper_parameters(key, prec):
    """The exact hyper-parameters of the reference's solo / duo models (hp.data of models/nucleiDAPI1-5 and
    models/nucleiDAPILAMIN: 64x64x1 with widths 80..1280, 128x128x2 with widths 36..1152).  Their weight shards are
    not shipped (reference Dockerfile:5-6), so seeded weights stand in: this pins shapes / planner / kernels at full width,
    not the trained numerics."""
    from oracle import oracle
    hp = model.KNOWN_HP[key]
    blob = model.random_blob(hp, seed=7)
    x = np.random.default_rng(3).normal(size=(3, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
    ref = oracle.forward(hp, blob, x)
    with umx.Engine(hp, blob, max_batch=2, precision=prec) as eng:
        got = eng.forward_tiles(x)
    assert np.abs(got - ref).max() <= TILE_TOL


@pytest.mark.parametrize("shape", [(5, 7), (31, 200), (129, 33)])
def test_infer_image_edge_sizes(shape):
    """Images smaller than one tile / one sub-patch and ragged sizes (the reference pads to whole sub-patches,
    PartitionOfImage.py:46-63): stitched planes vs the reference-equivalent oracle loop."""
    from oracle import oracle
    hp = helpers.small_hps()["v2_solo_like"]
    blob = model.random_blob(hp, seed=5)
    img = np.random.default_rng(11).random(shape) * 0.7
    with umx.Engine(hp, blob, max_batch=3) as eng:
        got = eng.infer_image(img, 0.3, 0.2)
    assert got.shape == (hp.nClasses,) + shape
    for k in range(hp.nClasses):
        ref = oracle.single_image_inference(hp, blob, img, 0.3, 0.2, "accumulate", k, duplicate_plane=False)
        assert np.abs(got[k].astype(np.float32) - ref.astype(np.float32)).max() <= 1e-3


def _random_hps(n, seed):
    """Seeded hyper-parameter sets across both graphs: filter sizes 3/5/7, 1-3 input channels, 2-4 classes, 1-5 levels
    (bottoms down to 2x2), 0-2 extra convs, widths that are not multiples of 8 or 16."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        graph = int(rng.integers(0, 2))
        L = int(rng.integers(1, 6))
        ks = int(rng.choice([3, 3, 5, 7]))
        P = int(rng.choice([16, 32, 64, 128]))
        if P >> L < 2 or (ks == 7 and P >> L < 4):
            continue
        hp = model.HParams(graph, P, int(rng.integers(1, 4)), int(rng.integers(2, 5)), int(rng.integers(3, 25)), L, ks,
                           int(rng.integers(0, 3)) if graph == model.GRAPH_LEGACY else int(rng.integers(0, 2)))
        if hp.flops_per_tile() > 3e9:
            continue
        out.append(hp)
    return out


@pytest.mark.parametrize("prec", PRECS + ["default"])
def test_forward_tiles_random_hyper_parameters(prec):
    """Fuzz of the planner: 14 seeded configurations, each either runs within tolerance or is refused at create time with
    UMX_ERR_INVALID (a geometry the kernels do not cover) -- never a wrong answer, never a fault.  The split-precision
    planner covers fewer tiny-layer geometries than the fp32 kernels; "default" picks whichever covers the graph."""
    from oracle import oracle
    ran = 0
    need = {"f32": 10, "default": 10, "f16x3": 8}[prec]   # (f16x3: 11 of 14 by the host-side planner check, tests/test_plan_check_cpu.py)
    for i, hp in enumerate(_random_hps(14, 2026)):
        blob = model.random_blob(hp, seed=100 + i)
        x = np.random.default_rng(i).normal(size=(3, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
        try:
            eng = umx.Engine(hp, blob, max_batch=2, precision=prec)
        except umx.UmxError as e:
            assert e.code == 1, (hp, e)
            continue
        with eng:
            got = eng.forward_tiles(x)
        ref = oracle.forward(hp, blob, x)
        assert np.abs(got - ref).max() <= TILE_TOL, (hp, np.abs(got - ref).max())
        ran += 1
    assert ran >= need
