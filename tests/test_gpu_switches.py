"""Every planner switch libumx still reads from the environment (and the `lanes` option), exercised on the GPU: the same tiles
and the same slide must come out within the parity tolerance of the oracle -- or equal to the default path, where the switch
changes scheduling only.  A switch without a test here does not exist in the library.

Round 5 retired the A/B switches whose answer is settled: UMX_PLANAR, UMX_XCD_ORDER, UMX_NO_{KSTEP_CARRY,
FUSED_HEAD, FUSED_CONVT, FIRST, FOLD, CONVT3, NT_TRIAL, D2S_SKIP, D2S, PACKED_TILE, RAW_GATHER, RAW_RESCALE}, UMX_HOST_SLABS.  The kernel
forms they selected are still reached by shape (tests/test_gpu_parity.py's graphs, D2S_SHAPES below: fused-phase, per-phase and
depth-to-space transposed convolutions; packed and plain last N-tiles; dense-K and generic first layers) and every one is held
to the oracle there.  What is left: precision, activation headroom, and the planner's debug overrides."""
import numpy as np
import pytest

from unmicst_amd import model, umx

pytestmark = pytest.mark.gpu

TILE_TOL = 1e-4

# a v2 graph large enough to have octet-planar tensors (>= 16 x 16 pixels, > 8 stored channels), fused-phase transposed
# convolutions of 2 and 4 N-tiles and a per-phase one (28 / 56 / 112 output channels), the dense-K first layer, the raw-skip
# fold with compact input tiles (28 real of 32 stored channels at the top), a fused head and k-step carries
HP = model.HParams(model.GRAPH_V2, 64, 2, 3, 28, 3, 3, 0, 2)

SWITCHES = [
    {},                                     # the default plan (reference for the scheduling-only switches)
    {"UMX_PLAN_OVERRIDE": "lu0.conv:3:2,ld1.conv:1:1:12"},   # forced (octets per chunk, k-steps per stage[, piece-index array])
    {"UMX_PLAN_NT": "ld0.conv:1"},          # a first layer in two N-blocks: no dense-K kernel -> the graph is rebuilt without the raw-skip fold (not failed into fp32)
    {"UMX_PLAN_NT": "lu2.convT:4"},         # forced N-tiles per workgroup of one layer (here 8 padded N-tiles in two blocks instead of 7 in one)
    {"UMX_PRECISION": "f32"},               # default precision from the environment
    {"UMX_ACT_SHIFT": "2"},                 # activations stored times 4
]


def _inputs():
    rng = np.random.default_rng(77)
    blob = model.random_blob(HP, seed=9)
    x = rng.normal(size=(5, HP.imSize, HP.imSize, HP.nChannels)).astype(np.float32)
    img = rng.random((2, 150, 210)) * 0.7
    return blob, x, img


@pytest.fixture(scope="module")
def reference():
    from oracle import oracle
    blob, x, img = _inputs()
    return blob, x, img, oracle.forward(HP, blob, x)


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_switch_keeps_parity(env, reference, monkeypatch):
    blob, x, img, ref = reference
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with umx.Engine(HP, blob, max_batch=3) as eng:
        got = eng.forward_tiles(x)
        planes = eng.infer_image(img, 0.3, 0.2)
    assert np.abs(got - ref).max() <= TILE_TOL, env
    for k in env:
        monkeypatch.delenv(k)
    with umx.Engine(HP, blob, max_batch=3) as eng:
        base = eng.infer_image(img, 0.3, 0.2)
    # whole image: <= 1e-3 = 2 fp16 ulp below 1.0 (tile probabilities move by ~1e-6 between plans, which can flip a rounding)
    assert np.abs(planes.astype(np.float32) - base.astype(np.float32)).max() <= 1e-3, env


def _kernels_run(eng, x, key="kernel"):
    eng.profile_enable(1)
    eng.forward_tiles(x)
    prof = eng.profile_read()
    eng.profile_enable(False)
    return {p["name"]: p[key] for p in prof if p["kernel"].startswith("conv_f16x3")}


def test_workgroup_order_2_is_chosen_by_rule(reference):
    """Workgroup order 2 ((N-block, phase) fastest inside an XCD) is taken per launch where the rule of run_launch_f16 asks for it
    -- plain convolutions of 2 - 4 N-blocks and halo-bound many-block layers on >= 64 tiles -- and nowhere on a handful of tiles;
    the profile entry carries the order a site's last launch used, and the results do not depend on it."""
    blob, x, img, ref = reference
    hp = model.HParams(model.GRAPH_V2, 32, 2, 3, 40, 3, 3, 0, 2)      # 160 / 320 output channels: layers of 2 and 3 N-blocks
    b2 = model.random_blob(hp, seed=2)
    few = np.random.default_rng(1).normal(size=(5, 32, 32, 2)).astype(np.float32)
    many = np.random.default_rng(1).normal(size=(320, 32, 32, 2)).astype(np.float32)    # 8 x 8-pixel layers: 4 images per workgroup tile
    many[:5] = few
    with umx.Engine(hp, b2, max_batch=5) as eng:
        small = eng.forward_tiles(few)
        assert 2 not in _kernels_run(eng, few, "xcd_order").values()     # (5 tiles: below the rule's 64)
    with umx.Engine(hp, b2, max_batch=320) as eng:
        big = eng.forward_tiles(many)
        orders = _kernels_run(eng, many, "xcd_order")
    assert sum(1 for o in orders.values() if o == 2) >= 2, orders
    assert np.array_equal(big[:5], small)


def test_two_lanes_option_equals_one_lane(reference):
    """umx_options.lanes = 2: tile batches alternate between two activation arenas on two streams; same launches, same bits."""
    blob, x, img, ref = reference
    with umx.Engine(HP, blob, max_batch=2, lanes=1) as e1, umx.Engine(HP, blob, max_batch=2, lanes=2) as e2:
        a, b = e1.forward_tiles(x), e2.forward_tiles(x)
        pa, pb = e1.infer_image(img, 0.3, 0.2), e2.infer_image(img, 0.3, 0.2)
    assert np.array_equal(a, b) and np.abs(a - ref).max() <= TILE_TOL
    assert np.array_equal(pa.view(np.uint16), pb.view(np.uint16))


def test_more_than_eight_input_channels(reference):
    """nChannels > 8: the input tiles span two octets and stay NHWC (the advisor's round-2 finding: they were addressed as
    octet-planar); both precisions against the oracle."""
    from oracle import oracle
    hp = model.HParams(model.GRAPH_V2, 32, 11, 3, 16, 2, 3, 0, 2)
    blob = model.random_blob(hp, seed=3)
    x = np.random.default_rng(4).normal(size=(3, 32, 32, 11)).astype(np.float32)
    want = oracle.forward(hp, blob, x)
    for prec in ("f16x3", "f32"):
        with umx.Engine(hp, blob, max_batch=3, precision=prec) as eng:
            got = eng.forward_tiles(x)
        assert np.abs(got - want).max() <= TILE_TOL, prec


# (nOut0, nChannels, ks, imSize): the depth-to-space shapes of make_d2s -- per transposed convolution F = Cout // 8 full octets
# and R = Cout % 8 left-over channels per phase -- one block of four phases or two blocks by output-row parity, with and
# without the remainder tile, with (1-2 input channels) and without (3) the raw-skip fold into the top layer's remainder tile
D2S_SHAPES = [
    (18, 2, 3, 16),    # 16-pixel tiles fold too (S2 >= 16), but neither the depth-to-space nor the fused-phase form takes a 16-pixel append: per-phase + append
    (20, 1, 3, 16),
    (18, 1, 3, 32),    # top 36 -> 18: F 2, R 2, one block of 5 tiles + fold (one compact channel); below 72 -> 36: F 4, R 4, 9 tiles
    (20, 2, 3, 64),    # 20: F 2, R 4 (5 tiles, fold of two channels); 40: F 5 -> two blocks of 5
    (22, 3, 3, 32),    # 22: R 6 -> stays on the fused-phase kernel; 44: F 5, R 4 -> two blocks of 6 with a two-phase remainder tile
    (24, 3, 3, 32),    # 24: F 3, R 0 (6 tiles, no remainder tile, no fold); 48: two blocks of 6
    (33, 2, 3, 32),    # 33: F 4, R 1 (9 tiles, fold); 66: F 8, R 2 -> two blocks of 9 with remainder tiles
    (36, 2, 3, 64),    # the duo widths: 36 (9 tiles, fold) and 72 (two blocks of 9)
    (40, 1, 3, 32),    # 40: F 5 -> two blocks of 5; 80 (the solo model's top layer, F 10) stays on the fused-phase kernel
    (12, 1, 5, 32),    # 5 x 5 filters: the phases have 9 / 6 / 6 / 4 taps in a 3 x 3 window; 24 -> 12: F 1 -> not eligible; 48 -> 24: 6 tiles
]


@pytest.mark.parametrize("shape", D2S_SHAPES, ids=lambda s: "n%d_c%d_k%d_s%d" % s)
def test_depth_to_space_transposed_convolution_shapes(shape):
    """conv_f16x3's D2S form against the oracle on graphs chosen to hit every branch of its N layout; the profile must show that the
    depth-to-space kernel actually ran."""
    from oracle import oracle
    n0, C, ks, S = shape
    hp = model.HParams(model.GRAPH_V2, S, C, 3, n0, 2, ks, 0, 2)
    blob = model.random_blob(hp, seed=40 + n0)
    x = np.random.default_rng(n0).normal(size=(5, S, S, C)).astype(np.float32)
    want = oracle.forward(hp, blob, x)
    with umx.Engine(hp, blob, max_batch=3, precision="f16x3") as eng:
        eng.profile_enable(1)
        got = eng.forward_tiles(x)
        kernels = {e["name"]: e["kernel"] for e in eng.profile_read()}
    assert np.abs(got - want).max() <= TILE_TOL, shape
    # (conv_f16x3<NT, KMT, NPH, DBG, MAXP, PK, D2S, F6, W2>: the seventh template argument)
    d2s = [n for n, k in kernels.items() if k.startswith("conv_f16x3<") and k[k.index("<") + 1:-1].split(", ")[6] == "true"]
    assert d2s and all("convT" in n for n in d2s), kernels


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8])
def test_raw_gather_equals_the_float64_image_path(dtype, reference):
    """The tile gather reads the raw planes and converts as it reads (im2double as one multiply, never fused with the
    normalisation; with `rescale` the drivers' rescale_intensity too): the same bytes as the float64 entry point fed with the
    host's own im2double [+ rescale] (toolbox/imtools.py:42-53, UnMicst.py:618-633)."""
    blob, x, img, ref = reference
    raw = np.random.default_rng(5).integers(0, np.iinfo(dtype).max + 1, size=(2, 150, 210)).astype(dtype)
    I = raw.astype(np.float64) * (1.0 / np.iinfo(dtype).max)
    R = np.stack([(np.clip(p, p.min(), p.max()) - p.min()) / (p.max() - p.min()) * 0.983 for p in I])
    with umx.Engine(HP, blob, max_batch=3) as eng:
        direct = eng.infer_image_raw(raw, False, 0.3, 0.2)
        rescaled = eng.infer_image_raw(raw, True, 0.3, 0.2)
        planes = eng.infer_image(I, 0.3, 0.2)
        planes_r = eng.infer_image(R, 0.3, 0.2)
    for got, pl in ((direct, planes), (rescaled, planes_r)):
        # the reference's double uint8 cast of the float16 planes (umx_kernels.hip half_to_u8_kernel)
        first = (np.float16(255) * pl).astype(np.uint8)
        want = (255.0 * (first.astype(np.float64) * (1.0 / 255))).astype(np.uint8)
        assert np.array_equal(got, want)


def test_the_stitch_kernels_double_to_half_equals_the_host_statement():
    """stitch_kernel converts with d2h_rne (round-to-odd binary32, then v_cvt_f16_f32); the unit test of the conversion only
    exercised the host routine.  Same directed vectors through the device routine: ties at every binade, half subnormals, values of the
    form 2^-25 (1 +- eps), the overflow boundary, signed zeros -- bit for bit the host routine's (and numpy's) rounding."""
    import numpy as np
    from unmicst_amd import umx
    rng = np.random.default_rng(0)
    eps = 2.0 ** -40
    vals = np.concatenate([
        rng.normal(size=100000), rng.normal(size=100000) * 1e-5, rng.normal(size=50000) * 1e-7, rng.uniform(0, 4, 100000),
        rng.normal(size=1000) * 7e4,
        np.array([0.0, -0.0, 1.0, 65504.0, 65519.99, 65520.0, 1e6, np.inf, -np.inf, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * (1 + eps),
                  2.0 ** -25 * (1 - eps), -2.0 ** -25 * (1 + eps), 3 * 2.0 ** -25, 3 * 2.0 ** -25 * (1 - eps), 2.0 ** -14,
                  2.0 ** -14 * (1 - 2.0 ** -12), 1 + 2.0 ** -11, 1 + 2.0 ** -11 + eps, 1 + 2.0 ** -11 - eps, 1 + 3 * 2.0 ** -11, 0.1, 1 / 3]),
        (np.arange(1024, 2048)[None, :] + 0.5).ravel() * 2.0 ** -10,                                   # exact ties, normal range
        (np.arange(0, 1024)[None, :] + 0.5).ravel() * 2.0 ** -24,                                      # exact ties among the subnormals
        ((np.arange(0, 1024)[None, :] + 0.5) * 2.0 ** -24).ravel() * (1 + eps)])                       # ... and just above them
    host = umx.double_to_half(vals)
    dev = umx.double_to_half_dev(vals)
    assert np.array_equal(dev.view(np.uint16), host.view(np.uint16))
    with np.errstate(over="ignore"):
        assert np.array_equal(dev.view(np.uint16), vals.astype(np.float16).view(np.uint16))
