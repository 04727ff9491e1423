"""GPU parity at the FULL sizes of BASELINE.json's configs (1024x1024x1 solo, 4096x4096x2 duo, 16384-wide synthetic-256
bands) through properties that do not need an oracle run of the whole slide:

* recomposition -- the whole-image entry point must equal "PI2D gather + normalise on the CPU (oracle) -> the tile
  forward of the same engine (parity-tested against the oracle per tile) -> the reference's stitch in numpy";
* partition of unity -- the class planes of the blended output sum to 1 (the blend weights are the same for every class);
* tile periodicity -- an image with period `sub` (tile stride) in both axes gives every interior tile the same input,
  so the stitched fp16 planes are bit-periodic in the interior;
* sharding invariance -- bands of patch rows + halo tiles reproduce the single pass bit for bit;
* spot checks of individual tiles of the big slide against the CPU oracle (tolerance 1e-4, north_star)."""
import numpy as np
import pytest

from unmicst_amd import model, umx

pytestmark = pytest.mark.gpu


def _slide(C, H, W, seed):
    rng = np.random.default_rng(seed)
    y = np.arange(H)[None, :, None]
    x = np.arange(W)[None, None, :]
    c = np.arange(C)[:, None, None]
    base = 0.22 + 0.18 * np.sin(y / 37.0 + c) * np.cos(x / 53.0) + 0.1 * np.sin((x + 2 * y) / 11.0 + 2 * c)
    return np.clip(base + 0.12 * rng.random((C, H, W)), 0.0, 0.983)


def _sum_to_one(planes, tol):
    s = planes.astype(np.float32).sum(0)
    assert np.abs(s - 1.0).max() <= tol, np.abs(s - 1.0).max()


def test_solo_1024_recomposition_and_spot_tiles():
    """BASELINE configs[1]: unmicst-solo hyper-parameters on a 1024 x 1024 1-channel image (484 tiles of 64 x 64)."""
    from oracle import oracle, pi2d_oracle
    hp = model.KNOWN_HP["nucleiDAPI1-5"]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = 0.34, 0.25
    img = _slide(1, 1024, 1024, 1)[0]
    with umx.Engine(hp, blob, max_batch=242) as eng:
        assert eng.tile_grid(1024, 1024)[:2] == (22, 22)
        full = eng.infer_image(img, mean, std)
        full32 = eng.infer_image(img, mean, std, stitch=umx.STITCH_FP32)
        pi = pi2d_oracle.PI2DOracle(img, hp.imSize, hp.margin, "accumulate")
        assert pi.num_patches == 484
        tiles = pi2d_oracle.normalised_batch(pi, 0, 484, 1, mean, std, True)
        probs = eng.forward_tiles(tiles)
    _sum_to_one(full32, 2e-6)
    _sum_to_one(full, 2e-3)                      # three fp16 roundings
    want = pi2d_oracle.stitch_all_classes((1024, 1024), hp.imSize, probs)
    # the whole-image kernel normalises in float64 on the device, this route rounds the tile to float32 first: tile
    # probabilities agree to ~1e-6, which can move an fp16 word by one ulp
    assert np.abs(full.astype(np.float32) - want.astype(np.float32)).max() <= 1e-3
    assert (full.view(np.uint16) == want.view(np.uint16)).mean() > 0.995
    for t in (0, 21, 230, 483):                  # corners (zero-padded canvas) and interior
        ref = oracle.forward(hp, blob, tiles[t:t + 1])
        assert np.abs(probs[t:t + 1] - ref).max() <= 1e-4


def test_duo_4096_periodicity_unity_and_sharding():
    """BASELINE configs[2]: unmicst-duo hyper-parameters on a 4096 x 4096 2-channel image (43 x 43 tiles of 128 x 128)."""
    import torch
    from oracle import oracle, pi2d_oracle
    from unmicst_amd import sharding
    hp = model.KNOWN_HP["nucleiDAPILAMIN"]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = 0.18, 0.17
    sub = hp.imSize - 2 * hp.margin            # 96
    cell = _slide(2, sub, sub, 3)
    reps = 4096 // sub + 1
    img = np.tile(cell, (1, reps, reps))[:, :4096, :4096].copy()      # period `sub` in both axes
    with umx.Engine(hp, blob, max_batch=256) as eng:
        assert eng.tile_grid(4096, 4096)[:2] == (43, 43)
        full = eng.infer_image(img, mean, std)
        full32 = eng.infer_image(img, mean, std, stitch=umx.STITCH_FP32)
        d_img = torch.from_numpy(img).cuda()
        parts = [sharding.infer_band_local(eng, d_img, mean, std, r, 4, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT)
                 for r in range(4)]
        got = np.concatenate([p.cpu().numpy() for p in parts], axis=1)
        pi = pi2d_oracle.PI2DOracle(img, hp.imSize, hp.margin, "accumulate")
        spot = pi2d_oracle.normalised_batch(pi, 43 * 20 + 7, 2, 2, mean, std, False)
        p_spot = eng.forward_tiles(spot)
    assert np.array_equal(got.view(np.uint16), full.view(np.uint16))           # 4 bands == single pass
    _sum_to_one(full32, 2e-6)
    # interior: every pixel at least one tile away from the canvas border, so all covering tiles see periodic input
    a = full[:, hp.imSize:4096 - hp.imSize - sub, hp.imSize:4096 - hp.imSize - sub].view(np.uint16)
    b = full[:, hp.imSize + sub:4096 - hp.imSize, hp.imSize + sub:4096 - hp.imSize].view(np.uint16)
    assert np.array_equal(a, b)
    assert np.abs(p_spot - oracle.forward(hp, blob, spot)).max() <= 1e-4


def test_synthetic256_band_of_the_16384_slide():
    """BASELINE configs[3] / the bench workload: one rank's 2048-row band of the 16384-wide slide (946 tiles of 256 x 256 x 2,
    the per-GPU share at N = 8), direct pass vs two-band sharded pass, partition of unity, spot tiles vs the oracle."""
    import torch
    from oracle import oracle, pi2d_oracle
    from unmicst_amd import sharding
    hp = model.KNOWN_HP["synthetic-256"]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = 0.18, 0.17
    H, W = 2048, 16384
    img = _slide(2, H, W, 5)
    with umx.Engine(hp, blob, max_batch=256) as eng:
        assert eng.tile_grid(H, W)[0] * eng.tile_grid(H, W)[1] == 946
        full = eng.infer_image(img, mean, std)
        d_img = torch.from_numpy(img).cuda()
        parts = [sharding.infer_band_local(eng, d_img, mean, std, r, 2, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT)
                 for r in range(2)]
        got = np.concatenate([p.cpu().numpy() for p in parts], axis=1)
        pi = pi2d_oracle.PI2DOracle(img[:, :512, :1024], hp.imSize, hp.margin, "accumulate")
        spot = pi2d_oracle.normalised_batch(pi, 1, 1, 2, mean, std, False)
        p_spot = eng.forward_tiles(spot)
    assert np.array_equal(got.view(np.uint16), full.view(np.uint16))
    _sum_to_one(full, 2e-3)
    assert np.abs(p_spot - oracle.forward(hp, blob, spot)).max() <= 1e-4


def test_synthetic256_full_16384_slide_in_eight_bands():
    """The north star's slide: 16384 x 16384 x 2, 86 x 86 = 7396 tiles of 256 x 256.  The single pass must be bit-periodic
    in the interior for a tile-periodic image, and the 8-rank band decomposition (what `bench.py --gpus 8` runs, here
    rank after rank on one GPU) must reproduce it bit for bit."""
    import torch
    from unmicst_amd import sharding
    hp = model.KNOWN_HP["synthetic-256"]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = 0.18, 0.17
    N = 16384
    sub = hp.imSize - 2 * hp.margin            # 192
    cell = _slide(2, sub, sub, 7)
    reps = N // sub + 1
    img = np.tile(cell, (1, reps, reps))[:, :N, :N].copy()
    with umx.Engine(hp, blob, max_batch=256) as eng:
        npr, npc = eng.tile_grid(N, N)[:2]
        assert (npr, npc) == (86, 86)
        full = eng.infer_image(img, mean, std)
        d_img = torch.from_numpy(img).cuda()
        del img
        row = 0
        for r in range(8):
            part = sharding.infer_band_local(eng, d_img, mean, std, r, 8, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT)
            part = part.cpu().numpy()
            assert np.array_equal(part.view(np.uint16), full[:, row:row + part.shape[1]].view(np.uint16)), r
            row += part.shape[1]
        assert row == N
    P = hp.imSize
    a = full[:, P:N - P - sub, P:N - P - sub].view(np.uint16)
    b = full[:, P + sub:N - P, P + sub:N - P].view(np.uint16)
    assert np.array_equal(a, b)
    s = full[:, ::7, ::5].astype(np.float32).sum(0)
    assert np.abs(s - 1.0).max() <= 2e-3


def test_solo_full_16384_slide_in_eight_bands():
    """BASELINE configs[3] as worded: unmicst-solo hyper-parameters on a 16384 x 16384 1-channel slide -- 342 x 342 =
    116 964 tiles of 64 x 64, a very different launch-count regime from the 256-px tile.  Tile-periodic input: the single
    pass must be bit-periodic in the interior, the 8-band decomposition (`bench.py --gpus 8 --workload solo-16384`, here
    band after band on one GPU) must reproduce it bit for bit, and spot tiles must match the CPU oracle."""
    import torch
    from oracle import oracle, pi2d_oracle
    from unmicst_amd import sharding
    hp = model.KNOWN_HP["nucleiDAPI1-5"]
    blob = model.random_blob(hp, seed=20260101)
    mean, std = 0.34, 0.25
    N = 16384
    sub = hp.imSize - 2 * hp.margin            # 48
    cell = _slide(1, sub, sub, 9)[0]
    reps = N // sub + 1
    img = np.tile(cell, (reps, reps))[:N, :N].copy()
    with umx.Engine(hp, blob, max_batch=1024) as eng:
        npr, npc = eng.tile_grid(N, N)[:2]
        assert (npr, npc) == (342, 342)
        full = eng.infer_image(img, mean, std)
        d_img = torch.from_numpy(img[None]).cuda()
        row = 0
        for r in range(8):
            part = sharding.infer_band_local(eng, d_img, mean, std, r, 8, umx.MODE_ACCUMULATE, umx.STITCH_FP16_COMPAT)
            part = part.cpu().numpy()
            assert np.array_equal(part.view(np.uint16), full[:, row:row + part.shape[1]].view(np.uint16)), r
            row += part.shape[1]
        assert row == N
        del d_img
        # interior tiles all see the same input: one oracle tile pins them
        pi = pi2d_oracle.PI2DOracle(img[:4 * sub + 2 * hp.margin, :4 * sub + 2 * hp.margin], hp.imSize, hp.margin, "accumulate")
        tiles = pi2d_oracle.normalised_batch(pi, 0, pi.num_patches, 1, mean, std, True)
        probs = eng.forward_tiles(tiles)
    ref = oracle.forward(hp, blob, tiles[[0, 5, 10]])
    assert np.abs(probs[[0, 5, 10]] - ref).max() <= 1e-4
    P = hp.imSize
    a = full[:, P:N - P - sub, P:N - P - sub].view(np.uint16)
    b = full[:, P + sub:N - P, P + sub:N - P].view(np.uint16)
    assert np.array_equal(a, b)
    s = full[:, ::7, ::5].astype(np.float32).sum(0)
    assert np.abs(s - 1.0).max() <= 2e-3
