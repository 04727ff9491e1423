"""CPU tests of the host-side mirror of the reference interface: wrapper -> script argv translation, per-tool
argparse surfaces, file-name / file-type switch, pre/post-processing helpers, converted model directories, and that
the facade fails loudly without a GPU (no CPU fallback)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import helpers
from unmicst_amd import driver, imtools, model, tiffio, umx
from unmicst_amd.unet2d import UNet2D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wrapper():
    spec = importlib.util.spec_from_file_location("unmicstWrapper", os.path.join(ROOT, "unmicstWrapper.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_wrapper_translates_one_based_flags():
    """reference unmicstWrapper.py:35-38,63-85: channel/classOrder/GPU shift to 0-based, tool -> script."""
    w = _wrapper()
    tool, argv = w.script_argv(w.parse(["img.ome.tif", "--tool", "unmicst-duo", "--channel", "3", "5", "--GPU", "2",
                                        "--classOrder", "1", "2", "3", "--stackOutput", "--outputPath", "out",
                                        "--scalingFactor", "0.5", "--model", "m", "--verbose"]))
    assert tool == "unmicst-duo"
    a = driver.build_parser(driver.TOOLS[tool]).parse_args(argv)
    assert a.channel == ["2", "4"] and a.GPU == 1 and a.classOrder == [0, 1, 2] and a.stackOutput and a.verbose
    assert a.outputPath == "out" and a.scalingFactor == 0.5 and a.model == "m" and a.mean == -1 and a.std == -1
    tool, argv = w.script_argv(w.parse(["x.tif"]))           # defaults: solo, channel 1 -> 0, GPU 0 -> -1 (auto)
    assert tool == "unmicst-solo"
    a = driver.build_parser(driver.TOOLS[tool]).parse_args(argv)
    assert a.channel == ["0"] and a.GPU == -1 and a.classOrder == -1 and not a.stackOutput
    assert a.model == "nucleiDAPI1-5" and a.outputPath is None
    tool, argv = w.script_argv(w.parse(["x.tif", "--tool", "unmicst-legacy", "--channel", "2", "9"]))
    a = driver.build_parser(driver.TOOLS[tool]).parse_args(argv)
    assert tool == "unmicst-legacy" and a.channel == 1 and a.model == "nucleiDAPI"   # legacy: a single int
    tool, _ = w.script_argv(w.parse(["x.tif", "--tool", "something-else"]))
    assert tool == "unmicst-solo"                                                      # reference's else branch


def test_script_surfaces_match_reference_defaults():
    """Defaults of the per-tool scripts (reference UnMicst1-5.py:715-731, UnMicst2.py:693-707, UnMicst.py:545-560,
    UnMicstCyto2.py:682-698)."""
    for tool, model_name, chan in (("unmicst-solo", "nucleiDAPI1-5", [0]), ("unmicst-duo", "nucleiDAPILAMIN", [0]),
                                   ("unmicst-legacy", "nucleiDAPI", 0), ("UnMicstCyto2", "nucleiDAPI", 0)):
        a = driver.build_parser(driver.TOOLS[tool]).parse_args(["p.tif"])
        assert (a.model, a.channel, a.classOrder, a.mean, a.std, a.scalingFactor, a.GPU, a.outlier) == \
            (model_name, chan, -1, -1, -1, 1, -1, -1)
        assert hasattr(a, "verbose") == (tool != "UnMicstCyto2")
    for script in ("UnMicst.py", "UnMicst1-5.py", "UnMicst2.py", "UnMicstCyto2.py", "unmicstWrapper.py"):
        assert os.path.exists(os.path.join(ROOT, script))


def test_file_name_switch():
    solo, duo = driver.TOOLS["unmicst-solo"], driver.TOOLS["unmicst-duo"]
    assert driver.split_name("exemplar-001-cycle6.ome.tif", solo) == ("exemplar-001-cycle6", "ome.tif")
    assert driver.split_name("a.b.c.tif", solo) == ("a.b.c", "tif")          # solo: text after the LAST dot
    assert driver.split_name("a.b.c.tif", duo) == ("a", "b.c.tif")           # others: after the FIRST dot
    assert driver.split_name("s.ome.tiff", duo) == ("s", "ome.tiff")
    with pytest.raises(NotImplementedError):
        driver.split_name("noext", solo)
    with pytest.raises(NotImplementedError):
        driver.read_plane("x.czi", "czi", 0, solo)
    with pytest.raises(NotImplementedError):
        driver.read_plane("x.ome.tiff", "ome.tiff", 0, driver.TOOLS["UnMicstCyto2"])   # cyto lists ome.tif / btf only


def test_preprocess_matches_legacy_recipe():
    raw = helpers.load_sample_105()[0]
    resized, rescaled = driver.preprocess(raw, 1, -1)
    assert resized.dtype == np.float64 and resized.shape == raw.shape
    assert np.array_equal(resized, raw * (1.0 / 65535))
    ref = helpers.legacy_preprocess(raw)
    assert np.abs(rescaled - ref).max() < 1e-15 and rescaled.min() == 0 and abs(rescaled.max() - 0.983) < 1e-15
    _, clipped = driver.preprocess(raw, 1, 99.0)                      # --outlier: percentile becomes the max
    assert clipped.max() <= 0.983 + 1e-12 and (clipped >= 0.983 - 1e-12).mean() > 0.005
    small, _ = driver.preprocess(raw, 0.5, -1)                        # anti-aliased bilinear, int() truncation
    assert small.shape == (416, 480) and 0 <= small.min() and small.max() <= resized.max()
    assert abs(small.mean() - resized.mean()) < 2e-3


def test_uint8_recipe_truncates_twice():
    pm = np.linspace(0, 1, 4096, dtype=np.float16).reshape(64, 64)
    out = imtools.to_uint8_via_resize(pm, (64, 64))
    first = np.uint8(255 * pm)
    again = np.uint8(255 * np.multiply(first, 1.0 / 255, dtype=np.float64))
    assert np.array_equal(out, again) and (out <= first).all() and (first - out).max() <= 1
    up = imtools.to_uint8_via_resize(pm, (128, 96))
    assert up.shape == (128, 96) and up.dtype == np.uint8


def test_converted_model_dir_roundtrip(tmp_path):
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    d = str(tmp_path / "nucleiDAPI")
    model.save_converted(model.ModelArtefacts(hp, blob, mean, std), d)
    art = model.load_model_dir(d)
    assert art.hp == hp and np.array_equal(art.blob, blob) and (art.mean, art.std) == (mean, std)
    assert model.detect_graph(d) == model.GRAPH_LEGACY


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not mounted")
def test_detects_graph_of_reference_model_dirs():
    assert model.detect_graph("/root/reference/models/nucleiDAPI") == model.GRAPH_LEGACY
    assert model.detect_graph("/root/reference/models/nucleiDAPI1-5") == model.GRAPH_V2
    assert model.detect_graph("/root/reference/models/nucleiDAPILAMIN") == model.GRAPH_V2
    with pytest.raises(FileNotFoundError):     # weight shard not shipped: loud failure, like tf's restore error
        model.load_model_dir("/root/reference/models/nucleiDAPI1-5")
    art = model.load_model_dir("/root/reference/models/nucleiDAPILAMIN", synthetic_if_missing=True)
    assert art.hp == model.KNOWN_HP["nucleiDAPILAMIN"] and (art.mean, art.std) == (0.18, 0.17)


SHIPPED = {"nucleiDAPI": "model.ckpt", "mousenucleiDAPI": "nuclei20x2bin1chan", "CytoplasmIncell": "model.ckpt"}


@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_shipped_converted_models_match_their_golden_copies(name):
    """models/<name>/umx_model.npz (what `--model <name>` loads, reference UnMicst.py:547,572) is the committed conversion:
    same blob, hyper-parameters and normalisation scalars as the golden copy the parity tests run on."""
    hp, blob, mean, std = helpers.load_nuclei_dapi(name)
    art = model.load_model_dir(os.path.join(helpers.ROOT, "models", name))
    assert art.hp == hp == model.KNOWN_HP[name] and (art.mean, art.std) == (mean, std)
    assert np.array_equal(art.blob, blob) and blob.size == sum(int(np.prod(s)) for _, s in model.tensor_specs(hp))
    assert hp.graph == model.GRAPH_LEGACY and np.isfinite(blob).all()


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not mounted")
@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_converter_reproduces_the_shipped_models_from_the_reference_checkpoints(name, tmp_path):
    """tools/convert_model.py on the reference's own model directories (TF tensor-bundle read without TensorFlow).  The
    mousenucleiDAPI directory keeps its weights under another prefix than the `model.ckpt` the reference restores
    (UnMicst.py:500-503; that shard is missing from its tree), with 16 features where hp.data says 20: the hyper-parameters
    follow the checkpoint that is converted."""
    sys.path.insert(0, os.path.join(helpers.ROOT, "tools"))
    import convert_model
    src = os.path.join("/root/reference/models", name)
    argv = [src, str(tmp_path)] + (["--prefix", SHIPPED[name]] if SHIPPED[name] != "model.ckpt" else [])
    out = convert_model.main(argv)
    hp, blob, mean, std = helpers.load_nuclei_dapi(name)
    with np.load(out) as z:
        assert np.array_equal(z["blob"], blob) and model.hparams_from_vector(z["hp"]) == hp
        assert (float(z["mean"]), float(z["std"])) == (mean, std)
    if name == "mousenucleiDAPI":
        assert model.load_pickle(os.path.join(src, "hp.data"))["nOut0"] == 20 and hp.nOut0 == 16
        with pytest.raises(FileNotFoundError):   # the directory's own model.ckpt has no shard: loud, like tf's restore error
            model.load_model_dir(src)


def test_facade_has_no_cpu_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    d = str(tmp_path / "m")
    model.save_converted(model.ModelArtefacts(hp, blob, mean, std), d)
    with pytest.raises(umx.UmxError) as e:
        UNet2D.singleImageInferenceSetup(d, 0, -1, -1)
    assert e.value.code == 3   # UMX_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        UNet2D.singleImageInferenceAll(np.zeros((8, 8)))


class _OracleEngine:
    """ORACLE-backed stand-in for umx.Engine (tests may use the oracle; the product never does): lets the driver's
    host logic -- page selection, pre-processing, class order, uint8 recipe, file names -- run without a GPU."""
    device = 0

    def __init__(self, hp, blob, device=0, max_batch=32):
        self.hp, self.blob = hp, blob

    def infer_image(self, image, mean, std, mode=0, stitch=0):
        from oracle import oracle
        return np.stack([oracle.single_image_inference(self.hp, self.blob, image, mean, std,
                                                       "replace" if mode else "accumulate", k)
                         for k in range(self.hp.nClasses)])

    def infer_image_raw(self, raw, rescale, mean, std, mode=0, value_range=None):
        """Host restatement of umx_infer_image_raw (the GPU test checks the real one against the general host recipe)."""
        planes = raw[None] if raw.ndim == 2 else raw
        if value_range is not None:   # what the driver hands in must be what the recipe finds itself
            assert [tuple(r) for r in value_range] == [(int(p.min()), int(p.max())) for p in planes]
        pre = [driver.preprocess(p, 1, -1)[1 if rescale else 0] for p in planes]
        image = np.stack(pre) if raw.ndim == 3 else pre[0]
        pm = self.infer_image(image, mean, std, mode)
        return np.stack([imtools.to_uint8_via_resize(pm[k], pm.shape[1:]) for k in range(pm.shape[0])])

    def infer_image_raw_scaled(self, raw, scaling, rescale, mean, std, mode=0):
        """Host restatement of umx_infer_image_raw_scaled: the drivers' general recipe (driver.preprocess / to_uint8_via_resize)."""
        planes = raw[None] if raw.ndim == 2 else raw
        pre = [driver.preprocess(p, scaling, -1)[1 if rescale else 0] for p in planes]
        image = np.stack(pre) if raw.ndim == 3 else pre[0]
        pm = self.infer_image(image, mean, std, mode)
        return np.stack([imtools.to_uint8_via_resize(pm[k], planes[0].shape) for k in range(pm.shape[0])])

    def infer_image_raw_outlier(self, raw, scaling, outlier, mean, std, mode=0):
        """Host restatement of umx_infer_image_raw_outlier (np.percentile in driver.preprocess)."""
        planes = raw[None] if raw.ndim == 2 else raw
        pre = [driver.preprocess(p, scaling, outlier)[1] for p in planes]
        image = np.stack(pre) if raw.ndim == 3 else pre[0]
        pm = self.infer_image(image, mean, std, mode)
        return np.stack([imtools.to_uint8_via_resize(pm[k], planes[0].shape) for k in range(pm.shape[0])])

    def close(self):
        pass


def test_driver_end_to_end_with_oracle_engine(tmp_path, monkeypatch):
    """unmicst-legacy --stackOutput on the reference's sample image, engine swapped for the oracle: the files the driver
    writes match the reference's bundled outputs (<= 1 LSB compute + <= 1 LSB of the reference's own double cast)."""
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    models = tmp_path / "models"
    model.save_converted(model.ModelArtefacts(hp, blob, mean, std), str(models / "nucleiDAPI"))
    raw, g_cont, g_raw, g_nuc = helpers.load_sample_105()
    crop = (slice(0, 300), slice(100, 420))    # a crop keeps the CPU suite fast; min/max of the crop drive the rescale
    reg = tmp_path / "ex" / "registration"
    os.makedirs(reg)
    tiffio.imsave(str(reg / "s.tif"), raw[crop])
    tiffio.imsave(str(reg / "s.1.tif"), raw[crop])
    monkeypatch.setenv("UMX_MODELS_DIR", str(models))
    monkeypatch.setattr(umx, "Engine", _OracleEngine)
    monkeypatch.setattr(umx, "pick_device_most_free_memory", lambda: 0)
    calls = []
    real = _OracleEngine.infer_image
    monkeypatch.setattr(_OracleEngine, "infer_image", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    assert driver.run("unmicst-legacy", [str(reg / "s.tif"), "--stackOutput"]) == 0
    assert len(calls) == 1                     # all three class planes come from ONE pass over the image
    out = tmp_path / "ex" / "probability_maps"
    stack = tiffio.imread_all(str(out / "s_Probabilities_1.tif"))
    prev = tiffio.imread_all(str(out / "qc" / "s_Preview_1.tif"))
    assert stack.shape == (3, 300, 320) and stack.dtype == np.uint8 and prev.shape == (2, 300, 320)
    assert np.array_equal(prev[0], stack[1])
    c = raw[crop].astype(np.float64) / 65535
    assert np.array_equal(prev[1], np.uint8(255 * (c / c.max())))
    # expectation: the oracle pipeline on the same crop with the reference's recipe
    from oracle import oracle
    I = helpers.legacy_preprocess(raw[crop])
    for page, k in enumerate((2, 1, 0)):
        pm = oracle.single_image_inference(hp, blob, I, mean, std, "accumulate", k)
        exp = np.uint8(255 * np.multiply(np.uint8(255 * pm), 1.0 / 255, dtype=np.float64))
        assert np.array_equal(stack[page], exp)
    # solo quirk: the network sees the UN-rescaled image (UnMicst1-5.py:816-821); duo duplicates a single channel
    seen = {}
    monkeypatch.setenv("UMX_NO_RAW_PATH", "1")          # general host-side recipe: the engine sees the float64 image
    monkeypatch.setattr(_OracleEngine, "infer_image",
                        lambda self, image, *a, **k: (seen.setdefault("img", np.array(image)),
                                                      np.zeros((3,) + image.shape[-2:], np.float16))[1])
    driver.run("unmicst-legacy", [str(reg / "s.tif"), "--model", str(models / "nucleiDAPI"), "--outputPath",
                                  str(tmp_path / "o2")])
    with pytest.raises(NotImplementedError):   # legacy takes the type from the FIRST dot: "1.tif" is not a known type
        driver.run("unmicst-legacy", [str(reg / "s.1.tif"), "--outputPath", str(tmp_path / "o2")])
    assert abs(seen.pop("img").max() - 0.983) < 1e-12
    driver.run("unmicst-solo", [str(reg / "s.1.tif"), "--model", str(models / "nucleiDAPI"), "--outputPath",
                                str(tmp_path / "o3")])
    assert np.array_equal(seen.pop("img"), raw[crop] * (1.0 / 65535))
    assert os.path.exists(tmp_path / "o3" / "s.1_ContoursPM_1.tif")    # solo: stem = text before the LAST dot
    # fast path (scalingFactor 1, default range): raw planes + the tool's rescale flag go to umx_infer_image_raw
    monkeypatch.delenv("UMX_NO_RAW_PATH")
    monkeypatch.setattr(_OracleEngine, "infer_image_raw",
                        lambda self, r, rescale, *a, **k: (seen.setdefault("raw", (np.array(r), rescale)),
                                                           np.zeros((3,) + r.shape[-2:], np.uint8))[1])
    driver.run("unmicst-solo", [str(reg / "s.1.tif"), "--model", str(models / "nucleiDAPI"), "--outputPath",
                                str(tmp_path / "o4")])
    r, flag = seen.pop("raw")
    assert r.dtype == np.uint16 and np.array_equal(r, raw[crop]) and flag is False
    driver.run("unmicst-legacy", [str(reg / "s.tif"), "--model", str(models / "nucleiDAPI"), "--outputPath",
                                  str(tmp_path / "o5")])
    assert seen.pop("raw")[1] is True
    driver.run("unmicst-legacy", [str(reg / "s.tif"), "--model", str(models / "nucleiDAPI"), "--outputPath",
                                  str(tmp_path / "o6"), "--scalingFactor", "0.5"])     # not the fast path
    assert "raw" not in seen and seen.pop("img").shape == (150, 160)


def test_preview_page_by_lookup_equals_the_float64_recipe():
    """driver.preview_u8: np.uint8(255 * (im2double(raw) / max)) through a table of the 2^16 (2^8) possible values."""
    from unmicst_amd import driver, imtools
    rng = np.random.default_rng(5)
    for dt, top in ((np.uint16, 51234), (np.uint8, 201), (np.uint16, 65535)):
        raw = rng.integers(0, top + 1, (300, 257)).astype(dt)
        raw[17, 5] = top
        rawI = imtools.im2double(raw)
        want = np.uint8(255 * (rawI / np.max(rawI)))
        assert np.array_equal(driver.preview_u8(raw), want)
    small = rng.integers(0, 4000, (10, 12)).astype(np.uint16)          # (below the table's break-even: the direct recipe)
    assert np.array_equal(driver.preview_u8(small), np.uint8(255 * (imtools.im2double(small) / np.max(imtools.im2double(small)))))


def test_plane_range_on_threads_equals_numpy():
    """driver.plane_range: row bands on threads for large planes, plain reductions for small ones."""
    rng = np.random.default_rng(3)
    big = rng.integers(7, 60000, size=(2100, 2048), dtype=np.uint16)
    big[2099, 2047] = 65535
    big[0, 0] = 3
    assert driver.plane_range(big) == (3, 65535)
    small = rng.integers(0, 255, size=(40, 50), dtype=np.uint8)
    assert driver.plane_range(small) == (int(small.min()), int(small.max()))
    assert np.array_equal(driver.preview_u8(big, 65535), driver.preview_u8(big))
