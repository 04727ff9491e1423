"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/umx.h declares, and its host-side
helpers agree with numpy / the Python model description.  No compute entry point is called without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers
from unmicst_amd import build, model, umx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return umx.load()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "umx.h")).read()
    declared = sorted(set(re.findall(r"UMX_API\s+[\w\s\*]+?\b(umx_\w+)\s*\(", header)))
    assert len(declared) >= 18
    assert sorted(umx.EXPORTS) == declared
    raw = ctypes.CDLL(build.lib_path())
    for name in declared:
        assert hasattr(raw, name), name


def test_exports_every_symbol_of_the_training_header(lib):
    from unmicst_amd import trainer
    header = open(os.path.join(ROOT, "include", "umx_train.h")).read()
    declared = sorted(set(re.findall(r"UMX_API\s+[\w\s\*]+?\b(umx_\w+)\s*\(", header)))
    assert len(declared) >= 12
    assert sorted(trainer.EXPORTS) == declared
    raw = ctypes.CDLL(build.lib_path())
    for name in declared:
        assert hasattr(raw, name), name
    assert ctypes.sizeof(trainer._TrainOptions) == 22 * 4 + 8 + 8 * 4     # umx_train_options
    for kind, py in (("solo", trainer.solo_options()), ("duo", trainer.duo_options())):
        nat = trainer.native_options(kind)
        for k, v in vars(py).items():
            assert getattr(nat, k) == pytest.approx(v, rel=1e-6), (kind, k)


def test_trainer_fails_loudly_without_a_gpu(lib):
    from unmicst_amd import trainer
    if umx.device_count() > 0:
        pytest.skip("a GPU is present")
    hp = helpers.small_hps()["v2_duo_like"]
    with pytest.raises(umx.UmxError) as e:
        trainer.Trainer(hp, model.random_blob(hp), batch=2)
    assert e.value.code == 3
    with pytest.raises(ValueError):
        trainer.Trainer(helpers.small_hps()["legacy_k5"], model.random_blob(helpers.small_hps()["legacy_k5"]))


def test_library_has_no_hip_runtime_dependency():
    """libumx must bind to the process's single HIP runtime at load time (see umx._bind_hip_runtime)."""
    import subprocess
    out = subprocess.run(["readelf", "-d", build.lib_path()], capture_output=True, text=True).stdout
    assert "amdhip64" not in out


def test_double_to_half_is_numpy_rounding(lib):
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.normal(size=100000), rng.normal(size=100000) * 1e-5, rng.normal(size=50000) * 1e-7,
        rng.uniform(0, 4, 100000), rng.normal(size=1000) * 7e4,
        np.array([0.0, -0.0, 1.0, 65504.0, 65519.99, 65520.0, 1e6, np.inf, -np.inf, 2.0 ** -24, 2.0 ** -25,
                  2.0 ** -25 * (1 + 2.0 ** -40), 2.0 ** -14, 2.0 ** -14 * (1 - 2.0 ** -12), 1 + 2.0 ** -11,
                  1 + 2.0 ** -11 + 2.0 ** -40, 1 + 3 * 2.0 ** -11, 0.1, 1 / 3]),
        # exact ties at every binade: k + 0.5 ulp
        (np.arange(1024, 2048)[None, :] + 0.5).ravel() * 2.0 ** -10])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16)
    got = umx.double_to_half(vals)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    nan = umx.double_to_half(np.array([np.nan]))
    assert np.isnan(nan[0])


def test_describe_matches_python_flop_model(lib):
    for name, hp in list(model.KNOWN_HP.items()) + list(helpers.small_hps().items()):
        d = umx.describe(hp)
        if hp.graph == model.GRAPH_LEGACY and hp.nExtraConvs > 0:
            # nothing is folded away: executed-unpadded == algorithmic as written
            assert d["flops_per_tile"] == pytest.approx(hp.flops_per_tile(), rel=1e-12), name
        else:
            assert d["flops_per_tile"] <= hp.flops_per_tile() * (1 + 1e-12), name
        assert d["executed_flops_per_tile"] >= d["flops_per_tile"]
    # SURVEY.md section 2.2 figures (FLOP/tile as written)
    assert model.KNOWN_HP["nucleiDAPI1-5"].flops_per_tile() == pytest.approx(5.210e9, rel=1e-3)
    assert model.KNOWN_HP["nucleiDAPILAMIN"].flops_per_tile() == pytest.approx(5.418e9, rel=1e-3)
    assert model.KNOWN_HP["nucleiDAPI"].flops_per_tile() == pytest.approx(1.815e9, rel=1e-3)
    assert model.KNOWN_HP["synthetic-256"].flops_per_tile() == pytest.approx(2.167e10, rel=1e-3)


def test_invalid_hyper_parameters_are_rejected(lib):
    bad = model.HParams(model.GRAPH_V2, 48, 1, 3, 8, 2, 3, 0)  # imSize not a power of two
    with pytest.raises(umx.UmxError) as e:
        umx.describe(bad)
    assert e.value.code == 1


def test_no_silent_cpu_fallback(lib):
    """Without a GPU, creating an engine must fail loudly (UMX_ERR_NO_DEVICE), never compute on the host."""
    if umx.device_count() > 0:
        pytest.skip("a GPU is present")
    hp = helpers.small_hps()["v2_solo_like"]
    with pytest.raises(umx.UmxError) as e:
        umx.Engine(hp, model.random_blob(hp))
    assert e.value.code == 3


def test_precision_options_are_validated_before_any_device_work(lib):
    """umx_create_opts: unknown precision / out-of-range act_shift are UMX_ERR_INVALID even without a GPU; both real
    precisions reach the device check (and fail with UMX_ERR_NO_DEVICE here -- there is no CPU fallback for either)."""
    hp = helpers.small_hps()["v2_solo_like"]
    blob = model.random_blob(hp)
    with pytest.raises(umx.UmxError) as e:
        umx.Engine(hp, blob, precision=7)
    assert e.value.code == 1
    with pytest.raises(umx.UmxError) as e:
        umx.Engine(hp, blob, precision="f16x3", act_shift=9)
    assert e.value.code == 1
    if umx.device_count() == 0:
        for prec in ("f32", "f16x3", "default"):
            with pytest.raises(umx.UmxError) as e:
                umx.Engine(hp, blob, precision=prec)
            assert e.value.code == 3
    assert ctypes.sizeof(umx._Options) == 64      # umx_options: 4 + 12 reserved int32
    # name[48] kernel[64] launches, 4 sums, launches_seen, (xcd_order, pers_grid) -- and the library reports the same size
    assert ctypes.sizeof(umx.ProfEntry) == 48 + 64 + 8 + 4 * 8 + 8 + 8 == lib.umx_prof_entry_size()
