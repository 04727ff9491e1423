"""GPU parity on the reference's two other models with weights in its tree: models/mousenucleiDAPI (checkpoint
nuclei20x2bin1chan: legacy graph, 3 x 3 kernels, 3 layers, 256-pixel tile) and models/CytoplasmIncell (legacy, 3 x 3, 2 classes),
loaded by `--model` (reference UnMicst.py:547,572) and restored at UnMicst.py:489-503.  These are the only TRAINED weights that
take the ks = 3 transposed-convolution crop (0 before / 1 after) and the split-precision range checks; the reference holds no
output of them, so the oracle (pinned by the nucleiDAPI goldens on the shared primitives) is the yardstick."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from unmicst_amd import driver, model, tiffio, umx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE_TOL = 1e-4
MODELS = ["mousenucleiDAPI", "CytoplasmIncell"]
PRECS = ["f32", "f16x3"]


def _sample():
    raw = helpers.load_sample_105()[0]
    return raw, helpers.legacy_preprocess(raw)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", MODELS)
def test_forward_tiles_real_weights(name, prec):
    """Per-tile probabilities on real 105.tif tiles: <= 1e-4 of the C oracle in both precisions; the split-precision engine
    keeps its precision (UMX_ERR_RANGE would have made the facade fall back -- here it would raise)."""
    from oracle import oracle, pi2d_oracle
    hp, blob, mean, std = helpers.load_nuclei_dapi(name)
    _, I = _sample()
    pi = pi2d_oracle.PI2DOracle(I, hp.imSize, hp.margin, "accumulate")
    t0 = pi.num_patches // 2 - 2
    x = pi2d_oracle.normalised_batch(pi, t0, 5, 1, mean, std, False)
    ref = oracle.forward(hp, blob, x)
    with umx.Engine(hp, blob, max_batch=3, precision=prec) as eng:   # 5 tiles through groups of 3 + 2
        assert eng.precision == prec
        got = eng.forward_tiles(x)
        eng.synchronize()                                            # surfaces UMX_ERR_RANGE
    assert got.shape == ref.shape == (5, hp.imSize, hp.imSize, hp.nClasses)
    assert np.abs(got - ref).max() <= TILE_TOL
    assert np.allclose(got.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", MODELS)
def test_whole_image_matches_the_oracle_loop(name, prec):
    """umx_infer_image (gather + normalise, UNet, fp16 stitch) on a crop of 105.tif that is not a multiple of the sub-patch
    against the reference-equivalent loop of the oracle, class by class."""
    from oracle import oracle
    hp, blob, mean, std = helpers.load_nuclei_dapi(name)
    _, I = _sample()
    crop = np.ascontiguousarray(I[100:100 + 2 * hp.imSize + 37, 50:50 + 2 * hp.imSize - 29])
    with umx.Engine(hp, blob, max_batch=8, precision=prec) as eng:
        got = eng.infer_image(crop, mean, std)
    assert got.shape == (hp.nClasses,) + crop.shape and got.dtype == np.float16
    for k in range(hp.nClasses):
        ref = oracle.single_image_inference(hp, blob, crop, mean, std, "accumulate", k, batch_size=8)
        assert np.abs(got[k].astype(np.float32) - ref.astype(np.float32)).max() <= 1e-3   # two fp16 ulp below 1.0
        assert (got[k].view(np.uint16) == ref.view(np.uint16)).mean() > 0.98


@pytest.fixture(scope="module")
def sample_file(tmp_path_factory):
    base = tmp_path_factory.mktemp("shipped")
    raw = helpers.load_sample_105()[0]
    reg = base / "exemplar" / "registration"
    os.makedirs(reg)
    tiffio.imsave(str(reg / "105.tif"), raw, append=False)
    return base, str(reg / "105.tif")


def _expected_u8(name, raw):
    """What the legacy driver writes for class k (reference UnMicst.py:618-633,651-656): rescale to (0, 0.983), one pass of the
    network, np.uint8(255 * pm), resize at the identity grid (u8 / 255), np.uint8(255 * .) -- with the ORACLE as the network."""
    from oracle import oracle, pi2d_oracle
    from unmicst_amd import imtools
    hp, blob, mean, std = helpers.load_nuclei_dapi(name)
    I = helpers.legacy_preprocess(raw)
    probs = oracle.tile_probs(hp, blob, I, mean, std, batch_size=16)
    planes = pi2d_oracle.stitch_all_classes(I.shape, hp.imSize, probs)
    return [imtools.to_uint8_via_resize(planes[k], raw.shape) for k in range(hp.nClasses)]


@pytest.mark.parametrize("name", MODELS)
def test_legacy_script_with_model_flag_end_to_end(name, sample_file):
    """`python UnMicst.py 105.tif --model <name> --stackOutput` out of the box (the converted weights come from
    <repo>/models/<name>/umx_model.npz): pages in reversed class order, the preview = [class of page 1, raw / max]
    (reference UnMicst.py:651-665); every page against the same recipe run on the oracle: one fp16 ulp of the stitched plane
    is one uint8 LSB after the first cast, and the reference's second cast np.uint8(255 * (v / 255)) maps some v to v - 1
    (UnMicst.py:651-656), so a pixel may differ by 2 -- the bound tests/test_gpu_cli.py uses for the nucleiDAPI files."""
    base, img = sample_file
    out = str(base / ("out_" + name))
    env = {k: v for k, v in os.environ.items() if k not in ("UMX_MODELS_DIR", "UMX_HIP_RUNTIME", "UMX_PRECISION")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "UnMicst.py"), img, "--model", name, "--stackOutput", "--outputPath", out],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "range exceeded" not in r.stdout       # (the facade's exact-fp32 retry on UMX_ERR_RANGE did not happen)
    hp = model.KNOWN_HP[name]
    raw = helpers.load_sample_105()[0]
    stack = tiffio.imread_all(os.path.join(out, "105_Probabilities_1.tif"))
    prev = tiffio.imread_all(os.path.join(out, "qc", "105_Preview_1.tif"))
    assert stack.shape == (hp.nClasses,) + raw.shape and stack.dtype == np.uint8 and prev.shape == (2,) + raw.shape
    want = _expected_u8(name, raw)
    for page, k in enumerate(range(hp.nClasses)[::-1]):
        d = np.abs(stack[page].astype(int) - want[k].astype(int))
        assert d.max() <= 2 and (d <= 1).mean() > 0.9999 and (d == 0).mean() > 0.97, (name, k, d.max(), (d == 0).mean())
    assert np.array_equal(prev[0], stack[1])
    assert np.array_equal(prev[1], helpers.load_sample_105()[2])   # raw / max preview plane: the reference's own bytes


def test_two_class_model_without_stack_output_fails_like_the_reference(sample_file):
    """CytoplasmIncell has two classes: without --stackOutput the reference writes the contours file and then indexes
    classOrder[2] of range(2) (UnMicst.py:667-674) -- IndexError.  Same here, after the same first file."""
    base, img = sample_file
    out = str(base / "out_two_class")
    os.environ["UMX_MODELS_DIR"] = os.path.join(ROOT, "models")
    try:
        with pytest.raises(IndexError):
            driver.run("unmicst-legacy", [img, "--model", "CytoplasmIncell", "--outputPath", out])
    finally:
        del os.environ["UMX_MODELS_DIR"]
    assert os.path.exists(os.path.join(out, "105_ContoursPM_1.tif"))
    assert not os.path.exists(os.path.join(out, "105_NucleiPM_1.tif"))
