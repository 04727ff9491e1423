"""Host logic of ``UNet2D.train`` (unmicst_amd/train_loop.py) on the CPU: the loop is driven with a stand-in for the
HIP trainer that evaluates the training oracle (test infrastructure), on a small synthetic data set written in the
reference's file convention.  The GPU run of the same loop is tests/test_gpu_train.py::test_train_loop_end_to_end."""
import os
import pickle

import numpy as np
import pytest

from unmicst_amd import model, tiffio, train_loop, trainer


class OracleTrainer:
    """Same surface as trainer.Trainer, arithmetic by oracle/train_oracle.py (float32 torch on the CPU)."""

    def __init__(self, hp, blob, opts, batch, device):
        import torch
        from oracle import train_oracle as to
        self.to, self.torch, self.hp, self.batch = to, torch, hp, batch
        kw = {k: getattr(opts, k) for k in ("lr0", "decay_steps", "decay_rate", "momentum", "beta1", "beta2", "adam_eps",
                                             "reg_kind", "reg_down", "reg_bottom", "reg_up", "reg_top", "clip_eps",
                                             "drop_down_step", "drop_bottom", "drop_up0", "drop_up_step", "bn_momentum",
                                             "seed")}
        self.o = to.TrainOptions(optimizer="adam", **kw)
        self.st = to.TrainState(blob)
        self.closed = False

    def step(self, data, labels, weights, apply_update=True):
        loss = self.to.train_step(self.hp, self.st, data, labels, weights, self.o, dtype=self.torch.float32)
        return loss, self.st.last["data_term"], self.st.last["reg"]

    def probs(self):
        return self.st.last["probs"]

    def eval(self, data):
        return self.to.inference_probs(self.hp, self.st.blob, data, dtype=self.torch.float32)

    def blob(self):
        return self.st.blob.astype(np.float32)

    @property
    def step_count(self):
        return self.st.step

    def close(self):
        self.closed = True


def write_dataset(root, n, P, C, n_aug, seed):
    """Blobs on a noisy background in the reference's convention: I%05d_Img.tif (n_aug x C uint16 pages), _Ant.tif
    (1 background, 2 contour, 3 nucleus), _wt.tif (contour emphasis in [0, 1])."""
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    yy, xx = np.mgrid[0:P, 0:P]
    for i in range(n):
        cy, cx, r = rng.uniform(8, P - 8, 2).tolist() + [rng.uniform(4, 7)]
        d = np.hypot(yy - cy, xx - cx)
        ant = np.where(d < r - 1, 3, np.where(d < r + 1, 2, 1)).astype(np.uint8)
        base = 0.2 + 0.5 * (d < r)
        path = os.path.join(root, "I%05d_Img.tif" % i)
        for page in range(max(n_aug, 1) * C):
            img = np.clip(base + rng.normal(0, 0.03, (P, P)), 0, 1)
            tiffio.imsave(path, np.uint16(img * 65535), append=page > 0)
        tiffio.imsave(os.path.join(root, "I%05d_Ant.tif" % i), ant)
        tiffio.imsave(os.path.join(root, "I%05d_wt.tif" % i), (ant == 2).astype(np.float32))


HP = {"imSize": 32, "nClasses": 3, "nChannels": 1, "nExtraConvs": 0, "nLayers": 2, "featMapsFact": 2, "downSampFact": 2,
      "ks": 3, "nOut0": 4, "stdDev0": 0.03, "batchSize": 2}


@pytest.mark.parametrize("regime", ["solo", "duo"])
def test_train_loop_files_and_bookkeeping(tmp_path, regime):
    rec = train_loop.RECIPES[regime]
    hp = dict(HP, nChannels=1 if regime == "solo" else 2)
    for name, n in (("train", 6), ("valid", 4), ("test", 3)):
        write_dataset(str(tmp_path / name), n, 32, hp["nChannels"], rec.n_aug, seed=hash(name) % 1000)
    np.random.seed(3)
    made = []

    def factory(hp_, blob_, opts_, batch_, device_):
        made.append(OracleTrainer(hp_, blob_, opts_, batch_, device_))
        return made[-1]

    hist = train_loop.train(hp, str(tmp_path / "train"), str(tmp_path / "valid"), str(tmp_path / "test"),
                            str(tmp_path / "log"), str(tmp_path / "model"), str(tmp_path / "pm"), 6, 4, 3, False, 3, 0, 2,
                            regime=regime, trainer_factory=factory)
    assert len(hist) == 3 and all(np.isfinite(l) for l, _ in hist)
    assert all(t.closed for t in made) and len(made) == 2            # training session + restored test session
    mdir = tmp_path / "model"
    assert pickle.load(open(mdir / "datasetMean.data", "rb")) == rec.dataset_mean
    assert pickle.load(open(mdir / "datasetStDev.data", "rb")) == rec.dataset_stdev
    assert pickle.load(open(mdir / "hp.data", "rb")) == hp
    art = model.load_model_dir(str(mdir))                                # step 0 always improves on inf: a model is saved
    assert art.hp.nOut0 == 4 and art.mean == rec.dataset_mean and art.blob.size == made[0].st.blob.size
    for sub in ("Train", "Valid"):
        rows = open(tmp_path / "log" / sub / "scalars.csv").read().strip().splitlines()
        assert rows[0].startswith("step,avg_cross_entropy,avg_pixel_error_0") and len(rows) >= 2
    names = sorted(os.listdir(tmp_path / "pm"))
    if regime == "solo":
        assert len(names) == 3 * 12 * 2 and names[0] == "I00001_0_Con.png"
    else:
        assert names == ["I00001Con.png", "I00001Nuc.png", "I00002Con.png", "I00002Nuc.png", "I00003Con.png", "I00003Nuc.png"]
    png = open(tmp_path / "pm" / names[0], "rb").read()
    assert png[:8] == b"\x89PNG\r\n\x1a\n" and png[16:24] == (96).to_bytes(4, "big") + (32).to_bytes(4, "big")


def test_load_split_follows_the_reference_convention(tmp_path):
    write_dataset(str(tmp_path), 2, 32, 1, 12, seed=5)
    hp = model.hparams_from_dict(HP, model.GRAPH_V2)
    rec = train_loop.RECIPES["solo"]
    X, L, W = train_loop.load_split(str(tmp_path), 2, [1, 0], hp, rec)
    assert X.shape == (2, 32, 32, 12, 1) and L.shape == W.shape == (2, 32, 32, 3)
    raw = tiffio.imread(str(tmp_path / "I00001_Img.tif"), key=7)
    assert np.allclose(X[0, :, :, 7, 0], (raw / 65535.0 - 0.34) / 0.25)
    ant = tiffio.imread(str(tmp_path / "I00001_Ant.tif"))
    assert np.array_equal(L[0].argmax(-1) + 1, ant) and np.all(L.sum(-1) == 1)
    assert set(np.unique(W[0, :, :, 0])) == {1.0} and set(np.unique(W[0, :, :, 2])) == {7.0}
    assert set(np.unique(W[0, :, :, 1])) == {2.0, 17.0}                 # contour_weight + intersect_weight * wt


def test_pixel_errors_and_initialiser():
    labels = np.eye(3)[np.array([[[0, 1], [2, 2]]])]
    probs = np.eye(3)[np.array([[[0, 2], [2, 1]]])].astype(np.float32)
    assert np.allclose(train_loop.pixel_errors(probs, labels), [0.0, 1.0, 0.5])
    hp = model.hparams_from_dict(dict(HP, nOut0=16), model.GRAPH_V2)
    T = model.tensors_from_blob(hp, train_loop.initial_blob(hp, 0.03, np.random.default_rng(0)))
    assert abs(T["ld0.w1"].std() - 0.03 * 0.8796) < 0.004 and np.abs(T["ld0.w1"]).max() <= 0.06 + 1e-6
    w = T["lu0.w2"]                                                       # [3,3,17,16]: fan_in 9*17, unit-variance scaling
    assert abs(w.std() - np.sqrt(1 / (9 * 17))) < 0.004
    assert np.all(T["lb.bn.gamma"] == 1) and np.all(T["lb.bn.var"] == 1) and not T["lb.bn.beta"].any()


def test_unet2d_facade_records_hyper_parameters():
    from unmicst_amd.unet2d import UNet2D
    UNet2D.setup(64, 1, 3, 80, 2, 2, 3, 0, 0.03, 4, 32)
    assert UNet2D.hp["nOut0"] == 80 and UNet2D.hp["stdDev0"] == 0.03 and UNet2D.hparams.nLayers == 4
    UNet2D.setupWithHP(dict(HP))
    assert UNet2D.hparams.imSize == 32 and trainer.solo_options().drop_bottom == 0.35


def test_deploy_writes_the_reference_files(tmp_path):
    """UNet2D.deploy host logic with an oracle-backed engine stand-in (the GPU run is test_gpu_train.py::test_deploy_on_the_gpu)."""
    from oracle import oracle

    class OracleEngine:
        def __init__(self, hp, blob, device, max_batch):
            self.hp, self.blob = hp, blob

        def forward_tiles(self, x):
            return oracle.forward(self.hp, self.blob, x)

        def close(self):
            pass

    hp = model.hparams_from_dict(dict(HP, nChannels=2), model.GRAPH_V2)
    model.save_converted(model.ModelArtefacts(hp, model.random_blob(hp, seed=3), 0.3, 0.2), str(tmp_path / "model"))
    write_dataset(str(tmp_path / "imgs"), 5, 32, 2, 0, seed=1)
    train_loop.deploy(str(tmp_path / "imgs"), 5, str(tmp_path / "model"), str(tmp_path / "pm"), 0, 2, engine_factory=OracleEngine)
    names = sorted(os.listdir(tmp_path / "pm"))
    assert names == ["I%05d_%s.png" % (i, t) for i in range(1, 6) for t in ("Im", "PM")]
    png = open(tmp_path / "pm" / "I00003_PM.png", "rb").read()
    assert png[:8] == b"\x89PNG\r\n\x1a\n" and png[16:24] == (32).to_bytes(4, "big") * 2
