"""GPU end-to-end tests of the reference-compatible entry points: unmicstWrapper.py / UnMicst.py run the shipped
nucleiDAPI weights on the reference's "UNet sample data" 105.tif and the written TIFFs are compared with the
reference's own bundled outputs (<= 1 uint8 LSB from the engine, + the <= 1 LSB the reference's double uint8 cast
itself loses -- UnMicst.py:651-656 -- which the bundled files, produced by batchUnMicst.py:551-587 without the
intermediate cast, do not contain)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from unmicst_amd import driver, model, tiffio
from unmicst_amd.unet2d import UNet2D

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def workspace(tmp_path_factory):
    base = tmp_path_factory.mktemp("cli")
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    models = base / "models"
    model.save_converted(model.ModelArtefacts(hp, blob, mean, std), str(models / "nucleiDAPI"))
    raw = helpers.load_sample_105()[0]
    reg = base / "exemplar" / "registration"
    os.makedirs(reg)
    # channel 2 (1-based) holds the DAPI plane; channel 1 is a decoy, so channel selection is exercised
    tiffio.imsave(str(reg / "105.ome.tif"), np.zeros_like(raw), append=False)
    tiffio.imsave(str(reg / "105.ome.tif"), raw, append=True)
    return base, str(models), str(reg / "105.ome.tif")


def _check_planes(cont, nuc, prev_raw):
    _, g_cont, g_raw, g_nuc = helpers.load_sample_105()
    for got, gold in ((cont, g_cont), (nuc, g_nuc)):
        d = np.abs(got.astype(int) - gold.astype(int))
        assert d.max() <= 2 and (d <= 1).mean() > 0.999 and (d == 0).mean() > 0.88, (d.max(), (d == 0).mean())
    assert np.array_equal(prev_raw, g_raw)     # raw/max preview plane: bit-exact


def test_legacy_script_in_process(workspace):
    base, models, img = workspace
    os.environ["UMX_MODELS_DIR"] = models
    try:
        out = str(base / "out_a")
        assert driver.run("unmicst-legacy", [img, "--channel", "1", "--outputPath", out]) == 0
    finally:
        del os.environ["UMX_MODELS_DIR"]
    cont = tiffio.imread_all(os.path.join(out, "105_ContoursPM_2.tif"))
    nuc = tiffio.imread_all(os.path.join(out, "105_NucleiPM_2.tif"))
    assert cont.shape == (2, 832, 960) and nuc.shape == (1, 832, 960) and cont.dtype == np.uint8
    assert os.path.isdir(os.path.join(out, "qc"))
    _check_planes(cont[0], nuc[0], cont[1])
    assert UNet2D.Engine is None               # cleanup ran


def test_wrapper_subprocess_stack_output(workspace):
    """The CI recipe of the reference (.github/workflows/ci.yml:34-35): wrapper, --stackOutput."""
    base, models, img = workspace
    env = dict(os.environ, UMX_MODELS_DIR=models, UMX_CLI_TIMING="1")
    env.pop("UMX_HIP_RUNTIME", None)     # the product default: the per-file tools bind the system HIP runtime and never import torch
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "unmicstWrapper.py"), "--tool", "unmicst-legacy",
                        "--channel", "2", "--stackOutput", "--outputPath", str(base / "out_b"), img],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "umx-cli-timing" in r.stderr
    assert not any(l.rstrip().endswith("| torch") for l in r.stderr.splitlines() if l.startswith("import time:")), "the CLI imported torch"
    stack = tiffio.imread_all(str(base / "out_b" / "105_Probabilities_2.tif"))
    prev = tiffio.imread_all(str(base / "out_b" / "qc" / "105_Preview_2.tif"))
    assert stack.shape == (3, 832, 960) and prev.shape == (2, 832, 960)
    # pages: class 2 (nuclei), 1 (contours), 0 (background) -- README.md:33, UnMicst.py:651
    _check_planes(stack[1], stack[0], prev[1])
    assert np.array_equal(prev[0], stack[1])
    total = stack.astype(int).sum(0)
    assert total.min() >= 250 and total.max() <= 256      # three truncated probabilities sum to ~255


def test_compat_per_class_writes_the_same_bytes(workspace):
    """--compat-per-class (the reference's loop: one whole pass per class, UnMicst.py:651-674): host-side
    pre-processing + one engine pass per class written, against the default single pass on the device.  Same files, byte for
    byte; the flag travels through the wrapper."""
    base, models, img = workspace
    os.environ["UMX_MODELS_DIR"] = models
    calls = []
    real = UNet2D.singleImageInferenceAll.__func__ if hasattr(UNet2D.singleImageInferenceAll, "__func__") else UNet2D.singleImageInferenceAll
    try:
        UNet2D.singleImageInferenceAll = staticmethod(lambda image, mode: (calls.append(1), real(image, mode))[1])
        a, b = str(base / "out_pc_a"), str(base / "out_pc_b")
        assert driver.run("unmicst-legacy", [img, "--channel", "1", "--stackOutput", "--outputPath", a]) == 0
        assert calls == []                                   # the default path never takes the per-class entry
        assert driver.run("unmicst-legacy", [img, "--channel", "1", "--stackOutput", "--outputPath", b, "--compat-per-class"]) == 0
        assert len(calls) == 3 and UNet2D.reuse_pass is True  # one whole pass per class; the switch is restored
    finally:
        UNet2D.singleImageInferenceAll = staticmethod(real)
        del os.environ["UMX_MODELS_DIR"]
    for name in ("105_Probabilities_2.tif", os.path.join("qc", "105_Preview_2.tif")):
        assert np.array_equal(tiffio.imread_all(os.path.join(a, name)), tiffio.imread_all(os.path.join(b, name))), name
    wrapper = __import__("importlib").util.spec_from_file_location("unmicstWrapper", os.path.join(ROOT, "unmicstWrapper.py"))
    m = __import__("importlib").util.module_from_spec(wrapper)
    wrapper.loader.exec_module(m)
    tool, argv = m.script_argv(m.parse(["--tool", "unmicst-legacy", "--compat-per-class", img]))
    assert tool == "unmicst-legacy" and argv[-1] == "--compat-per-class"
    assert "--compat-per-class" not in m.script_argv(m.parse(["--tool", "unmicst-legacy", img]))[1]


def test_default_output_dir_and_scaling(workspace):
    base, models, img = workspace
    os.environ["UMX_MODELS_DIR"] = models
    try:
        assert driver.run("unmicst-legacy", [img, "--channel", "1", "--scalingFactor", "0.5", "--mean", "0.2",
                                             "--std", "0.16", "--classOrder", "0", "2", "1"]) == 0
    finally:
        del os.environ["UMX_MODELS_DIR"]
    out = str(base / "exemplar" / "probability_maps")    # <parent of parent>/probability_maps (UnMicst.py:637-638)
    cont = tiffio.imread_all(os.path.join(out, "105_ContoursPM_2.tif"))
    assert cont.shape == (2, 832, 960)                   # resized back to the raw size
    nuc = tiffio.imread_all(os.path.join(out, "105_NucleiPM_2.tif"))[0]
    g_nuc = helpers.load_sample_105()[3]
    # classOrder 0 2 1 swaps the roles: "_ContoursPM_" now holds class 2; half-resolution inference of a model trained
    # at full resolution is only loosely comparable with the golden, so this is a plumbing check
    assert np.corrcoef(cont[0].ravel().astype(float), g_nuc.ravel().astype(float))[0, 1] > 0.5
    assert nuc.shape == (832, 960)


def test_raw_fast_path_is_bit_identical_to_the_host_recipe(workspace, monkeypatch):
    """umx_infer_image_raw (im2double, min/max, rescale_intensity, inference, double uint8 cast on the GPU) against the
    general host-side recipe of the driver, for the rescaled (legacy/duo) and un-rescaled (solo quirk) inputs, uint16 and
    uint8 planes."""
    from unmicst_amd import imtools, umx
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw16 = helpers.load_sample_105()[0][:400, :530]
    raw8 = np.uint8(raw16 >> 8)
    with umx.Engine(hp, blob, max_batch=32) as eng:
        for raw in (raw16, raw8):
            for rescale in (True, False):
                got = eng.infer_image_raw(raw, rescale, mean, std)
                resized, rescaled = driver.preprocess(raw, 1, -1)
                pm = eng.infer_image(rescaled if rescale else resized, mean, std)
                want = np.stack([imtools.to_uint8_via_resize(pm[k], raw.shape) for k in range(hp.nClasses)])
                assert got.dtype == np.uint8 and np.array_equal(got, want), (raw.dtype, rescale)
                if rescale:   # the extrema handed in (umx_infer_image_raw_range: slab-wise upload under the tile kernels)
                    ranged = eng.infer_image_raw(raw, True, mean, std, value_range=[driver.plane_range(raw)])
                    assert np.array_equal(ranged, want), raw.dtype
        with pytest.raises(umx.UmxError):   # not a (min, max) of 8-bit samples
            eng.infer_image_raw(raw8, True, mean, std, value_range=[(0, 300)])
        # two planes (duo-style input) are rescaled independently
        hp2 = helpers.small_hps()["v2_duo_like"]
    blob2 = model.random_blob(hp2, seed=8)
    two = np.stack([raw16[:90, :70], raw16[100:190, 200:270] // 3])
    with umx.Engine(hp2, blob2, max_batch=8) as eng:
        got = eng.infer_image_raw(two, True, 0.2, 0.2)
        pre = np.stack([driver.preprocess(p, 1, -1)[1] for p in two])
        pm = eng.infer_image(pre, 0.2, 0.2)
        want = np.stack([imtools.to_uint8_via_resize(pm[k], two.shape[1:]) for k in range(hp2.nClasses)])
        assert np.array_equal(got, want)
        assert np.array_equal(eng.infer_image_raw(two, True, 0.2, 0.2, value_range=[driver.plane_range(p) for p in two]), want)
    # a slide of several launch groups: the plain call finds the range on host threads under its own uploads, UMX_HOST_RANGE=0 on
    # the device behind the whole upload, the ranged call is handed it -- one result
    big = np.tile(raw16, (3, 2))[:1100, :1000]
    with umx.Engine(hp, blob, max_batch=16) as eng:
        plain = eng.infer_image_raw(big, True, mean, std)
        assert np.array_equal(eng.infer_image_raw(big, True, mean, std, value_range=[driver.plane_range(big)]), plain)
        monkeypatch.setenv("UMX_HOST_RANGE", "0")
        assert np.array_equal(eng.infer_image_raw(big, True, mean, std), plain)
        monkeypatch.delenv("UMX_HOST_RANGE")
        # ... and the rescale inside the raw tile gather against the float64 image the host recipe makes (min == max: the np.clip
        # branch of rescale_intensity)
        flat = np.full((300, 280), 1234, np.uint16)
        pre = driver.preprocess(flat, 1, -1)[1]
        pm = eng.infer_image(pre, mean, std)
        want_flat = np.stack([imtools.to_uint8_via_resize(pm[k], flat.shape) for k in range(hp.nClasses)])
        assert np.array_equal(eng.infer_image_raw(flat, True, mean, std), want_flat)
        pre_big = driver.preprocess(big, 1, -1)[1]
        pm = eng.infer_image(pre_big, mean, std)
        assert np.array_equal(plain, np.stack([imtools.to_uint8_via_resize(pm[k], big.shape) for k in range(hp.nClasses)]))


def test_two_slides_in_flight_equal_the_synchronous_calls(workspace):
    """umx_infer_image_raw_submit / umx_infer_image_wait: two slides enqueued on the two slots before either is waited for, with and
    without the intensity rescale (a submitted rescaled call finds its range on the device, the synchronous one on host threads), equal
    the synchronous calls byte for byte; a slot that still holds a call refuses the next one."""
    from unmicst_amd import umx
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw16 = helpers.load_sample_105()[0]
    a = np.ascontiguousarray(np.tile(raw16, (2, 2))[:900, :1100])
    b = np.ascontiguousarray(a[::-1, ::-1] // 2)
    with umx.Engine(hp, blob, max_batch=16) as eng:
        for rescale in (False, True):
            want = [eng.infer_image_raw(x, rescale, mean, std) for x in (a, b)]
            outs = [np.empty_like(want[0]), np.empty_like(want[1])]
            for slot, x in enumerate((a, b)):
                eng.infer_image_raw_submit(slot, x.ctypes.data, 16, 1, x.shape[0], x.shape[1], rescale, mean, std, outs[slot].ctypes.data)
            with pytest.raises(umx.UmxError):
                eng.infer_image_raw_submit(0, a.ctypes.data, 16, 1, a.shape[0], a.shape[1], rescale, mean, std, outs[0].ctypes.data)
            for slot in (1, 0):
                eng.infer_image_wait(slot)
            assert np.array_equal(outs[0], want[0]) and np.array_equal(outs[1], want[1]), rescale


def test_clean_checkout_runs_on_the_shipped_models_directory(workspace):
    """`python unmicstWrapper.py --tool unmicst-legacy img.tif` out of the box: no UMX_MODELS_DIR, the converted nucleiDAPI
    weights come from <repo>/models/nucleiDAPI/umx_model.npz (reference UnMicst.py:547,556: models/<--model>)."""
    base, _, img = workspace
    out = str(base / "out_clean")
    env = {k: v for k, v in os.environ.items() if k not in ("UMX_MODELS_DIR", "UMX_HIP_RUNTIME")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "unmicstWrapper.py"), "--tool", "unmicst-legacy", img, "--channel", "2",
                        "--outputPath", out], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    cont = tiffio.imread_all(os.path.join(out, "105_ContoursPM_2.tif"))
    nuc = tiffio.imread_all(os.path.join(out, "105_NucleiPM_2.tif"))
    _check_planes(cont[0], nuc[0], cont[1])


def test_duo_script_on_a_two_page_ome_tiff_4096(tmp_path):
    """BASELINE configs[2] file to file: UnMicst2.py (unmicst-duo, nucleiDAPILAMIN hyper-parameters from the shipped
    models/ directory, seeded synthetic weights -- the reference's shard is not in its tree) on a synthetic 4096 x 4096
    2-page OME-TIFF with --stackOutput; the written pages must equal what the facade returns for the same two planes
    (reference UnMicst2.py:747-751 channel pair, :780-788 rescale, :813-817 uint8 recipe, pages nuclei / contours /
    background)."""
    from unmicst_amd import imtools
    rng = np.random.default_rng(11)
    N = 4096
    y = np.arange(N)[:, None]
    x = np.arange(N)[None, :]
    planes = []
    for c in range(2):
        v = 0.25 + 0.2 * np.sin(y / (31.0 + 7 * c)) * np.cos(x / 47.0) + 0.1 * rng.random((N, N))
        planes.append(np.uint16(np.clip(v, 0, 1) * 60000))
    img = str(tmp_path / "slide.ome.tif")
    tiffio.imsave(img, planes[0], append=False)
    tiffio.imsave(img, planes[1], append=True)
    out = str(tmp_path / "out")
    env = {k: v for k, v in os.environ.items() if k not in ("UMX_MODELS_DIR", "UMX_HIP_RUNTIME")}
    env["UMX_SYNTHETIC_WEIGHTS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "UnMicst2.py"), img, "--channel", "0", "1", "--stackOutput",
                        "--outputPath", out], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = tiffio.imread_all(os.path.join(out, "slide_Probabilities_1.tif"))
    assert got.shape == (3, N, N) and got.dtype == np.uint8
    prev = tiffio.imread_all(os.path.join(out, "qc", "slide_Preview_1.tif"))
    assert prev.shape == (2, N, N)
    # the same planes through the facade (host-side recipe: resize at the identity grid, rescale to (0, 0.983))
    art = model.load_model_dir(os.path.join(ROOT, "models", "nucleiDAPILAMIN"), synthetic_if_missing=True)
    UNet2D.setupWithArtefacts(art)
    try:
        cells = np.stack([driver.preprocess(p, 1, -1)[1] for p in planes])
        want = [imtools.to_uint8_via_resize(UNet2D.singleImageInference(cells, "accumulate", k), (N, N)) for k in (2, 1, 0)]
    finally:
        UNet2D.singleImageInferenceCleanup()
    for page in range(3):
        assert np.array_equal(got[page], want[page]), page
    assert np.array_equal(prev[0], want[1])


@pytest.mark.parametrize("scaling,rescale", [(0.5, True), (0.75, False), (1.6, True)])
def test_scaled_raw_path_matches_the_host_recipe(scaling, rescale):
    """--scalingFactor != 1 on the device (umx_infer_image_raw_scaled: skimage.transform.resize's defaults restated in
    float64 kernels, both ways) against the host-side recipe (unmicst_amd/imtools.py: scipy.ndimage gaussian_filter + zoom):
    the resized input agrees to ~1e-15, so the uint8 planes differ at most where a value sits on a truncation boundary or a
    tile probability moves an fp16 ulp -- <= 1 LSB, and almost nowhere."""
    from unmicst_amd import imtools
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw = helpers.load_sample_105()[0][:400, :520]
    UNet2D.setupWithArtefacts(model.ModelArtefacts(hp, blob, mean, std))
    try:
        got = UNet2D.singleImageInferenceRawScaled(raw, scaling, rescale)
        pre = driver.preprocess(raw, scaling, -1)[1 if rescale else 0]
        want = np.stack([imtools.to_uint8_via_resize(UNet2D.singleImageInference(pre, "accumulate", k), raw.shape)
                         for k in range(hp.nClasses)])
    finally:
        UNet2D.singleImageInferenceCleanup()
    assert got.shape == want.shape == (hp.nClasses,) + raw.shape and got.dtype == np.uint8
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.995, (d.max(), (d == 0).mean())


@pytest.mark.parametrize("scaling,outlier", [(1, 99.0), (1, 37.5), (0.5, 99.9), (1.6, 50.0)])
def test_outlier_percentile_on_the_device_matches_numpy(scaling, outlier):
    """--outlier on the device (umx_infer_image_raw_outlier: the two order statistics by radix selection over the float64
    plane + numpy's linear interpolation) against the host recipe (np.percentile in driver.preprocess, reference
    UnMicst1-5.py:817-821).  At --scalingFactor 1 the rescaled planes are the same doubles, so the uint8 planes are equal; with
    a resize in front they inherit that path's <= 1 LSB."""
    from unmicst_amd import imtools
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw = helpers.load_sample_105()[0][:380, :500]
    UNet2D.setupWithArtefacts(model.ModelArtefacts(hp, blob, mean, std))
    try:
        got = UNet2D.singleImageInferenceRawOutlier(raw, scaling, outlier)
        pre = driver.preprocess(raw, scaling, outlier)[1]
        want = np.stack([imtools.to_uint8_via_resize(UNet2D.singleImageInference(pre, "accumulate", k), raw.shape)
                         for k in range(hp.nClasses)])
    finally:
        UNet2D.singleImageInferenceCleanup()
    assert got.shape == want.shape == (hp.nClasses,) + raw.shape and got.dtype == np.uint8
    d = np.abs(got.astype(int) - want.astype(int))
    if scaling == 1:
        assert d.max() == 0, (d.max(), (d == 0).mean())
    else:
        assert d.max() <= 1 and (d == 0).mean() > 0.995, (d.max(), (d == 0).mean())

