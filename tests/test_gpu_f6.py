"""UMX_PREC_F16X3_F6 (what UMX_PREC_DEFAULT selects) on the GPU: the cross terms of the wide plain convolutions at <= 1/4 resolution on the
block-scaled fp6 matrix instruction (conv_f16x3's F6 form: eight-wave workgroups over two tiles).  Same tolerance as every other
precision: 1e-4 per tile against the oracle (north_star); the CPU emulation of the arithmetic (tests/fp8_cross_term_report.py) puts it
at 3e-6 on these graphs."""
import numpy as np
import pytest

import helpers
from unmicst_amd import model, umx

pytestmark = pytest.mark.gpu
TILE_TOL = 1e-4
# a small graph with one layer the form takes: lb = 144 -> 288 channels (two blocks of nine N-tiles) at 16 x 16 pixels = 1/4 of the tile
HP_F6 = model.HParams(model.GRAPH_V2, 64, 2, 3, 72, 2, 3, 0)


@pytest.mark.parametrize("name,n", [("v2_72", 5), ("duo", 3)])
def test_forward_tiles_f16f6_matches_oracle(name, n):
    from oracle import oracle
    hp = HP_F6 if name == "v2_72" else model.KNOWN_HP["nucleiDAPILAMIN"]
    for seed in (11, 12):
        blob = model.random_blob(hp, seed=seed)
        x = np.random.default_rng(seed).normal(size=(n, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)   # (odd counts: a half-empty workgroup)
        ref = oracle.forward(hp, blob, x)
        with umx.Engine(hp, blob, max_batch=4, precision="f16f6") as eng:
            assert eng.precision == "f16f6"
            got = eng.forward_tiles(x)
            prof_ok = True
        err = float(np.abs(got - ref).max())
        assert err <= TILE_TOL, (name, seed, err)
        with umx.Engine(hp, blob, max_batch=4, precision="f16x3") as eng:
            base = eng.forward_tiles(x)
        assert not np.array_equal(got, base), "the fp6 form was not used on any layer"   # (it changes low-order bits where it runs)
        assert prof_ok


def test_f16f6_whole_image_equals_its_own_banded_run():
    """The decomposition invariants hold in this precision too: a slide computed in two launch-group sizes gives identical bytes
    (a tile's result must not depend on which tiles share its launch -- here: on its partner tile in the eight-wave workgroup)."""
    hp = HP_F6
    blob = model.random_blob(hp, seed=3)
    img = np.random.default_rng(5).random((2, 150, 170)) * 0.6
    outs = []
    for mb in (3, 8):
        with umx.Engine(hp, blob, max_batch=mb, precision="f16f6") as eng:
            outs.append(eng.infer_image(img, 0.2, 0.2))
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))


def test_the_default_precision_is_this_form_where_a_layer_takes_it():
    blob = model.random_blob(HP_F6, seed=2)
    with umx.Engine(HP_F6, blob, max_batch=2) as eng:
        assert eng.precision == "f16f6"
    hp = helpers.small_hps()["v2_wide"]          # 144 / 288 channels, but no plain convolution of two N-blocks at <= 1/4 resolution
    with umx.Engine(hp, model.random_blob(hp, seed=2), max_batch=2) as eng:
        assert eng.precision == "f16x3"


def test_models_without_wide_deep_layers_are_unchanged():
    hp = helpers.small_hps()["v2_duo_like"]
    blob = model.random_blob(hp, seed=4)
    x = np.random.default_rng(1).normal(size=(3, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
    res = []
    for prec in ("f16x3", "f16f6"):
        with umx.Engine(hp, blob, max_batch=4, precision=prec) as eng:
            res.append(eng.forward_tiles(x))
    assert np.array_equal(res[0], res[1])
