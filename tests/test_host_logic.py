"""CPU tests of the host logic: checkpoint reader, model artefacts, TIFF io, band partitioning."""
import os
import pickle
import struct

import numpy as np
import pytest

import helpers
from unmicst_amd import model, sharding, tfckpt, tiffio

REF = "/root/reference"


def _varint(n):
    out = b""
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def _write_bundle(prefix, tensors):
    """Write a minimal TF tensor-bundle (uncompressed LevelDB table, one data block) for the reader test."""
    data = b""
    entries = []
    for name in sorted(tensors):
        t = np.asarray(tensors[name], order="C")
        dtype = {np.dtype("float32"): 1, np.dtype("int32"): 3}[t.dtype]
        shape = b"".join(b"\x12" + _varint(len(d)) + d for d in [b"\x08" + _varint(s) for s in t.shape])
        msg = b"\x08" + _varint(dtype) + b"\x12" + _varint(len(shape)) + shape + b"\x20" + _varint(len(data)) + \
            b"\x28" + _varint(t.nbytes) + b"\x35" + struct.pack("<I", 0)
        entries.append((name.encode(), msg))
        data += t.tobytes()
    entries.insert(0, (b"", b"\x08\x01"))  # header entry under the empty key

    def block(kvs, prefix_compress):
        out, restarts, last = b"", [], b""
        for i, (k, v) in enumerate(kvs):
            shared = 0
            if prefix_compress and i % 4:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            else:
                restarts.append(len(out))
            out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
            last = k
        for r in restarts:
            out += struct.pack("<I", r)
        return out + struct.pack("<I", len(restarts))

    db = block(entries, True)
    buf = db + b"\0" + b"\0\0\0\0"
    meta_off = len(buf)
    mb = block([], False)
    buf += mb + b"\0" + b"\0\0\0\0"
    idx_off = len(buf)
    ib = block([(b"\xff", _varint(0) + _varint(len(db)))], False)
    buf += ib + b"\0" + b"\0\0\0\0"
    footer = _varint(meta_off) + _varint(len(mb)) + _varint(idx_off) + _varint(len(ib))
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    open(prefix + ".index", "wb").write(buf + footer)
    open(prefix + ".data-00000-of-00001", "wb").write(data)


def test_tensor_bundle_reader_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"downsampling/ld0/kernel1": rng.normal(size=(5, 5, 1, 16)).astype(np.float32),
               "downsampling/ld0/kernelExtra0": rng.normal(size=(5, 5, 16, 16)).astype(np.float32),
               "lt/kernel": rng.normal(size=(1, 1, 16, 3)).astype(np.float32),
               "Variable": np.array(7, dtype=np.int32)}
    prefix = str(tmp_path / "model.ckpt")
    _write_bundle(prefix, tensors)
    got = tfckpt.load_checkpoint(prefix)
    assert sorted(got) == sorted(tensors)
    for k in tensors:
        assert np.array_equal(got[k], tensors[k]) and got[k].shape == tensors[k].shape
    with pytest.raises(FileNotFoundError):
        tfckpt.load_checkpoint(str(tmp_path / "absent"))


@pytest.mark.parametrize("name", sorted(helpers.small_hps()))
def test_model_dir_roundtrip(tmp_path, name):
    """hp.data / datasetMean / datasetStDev / model.ckpt in the reference's layout -> the same canonical blob."""
    hp = helpers.small_hps()[name]
    blob = model.random_blob(hp, seed=2)
    tensors = model.tensors_from_blob(hp, blob)
    _write_bundle(str(tmp_path / "model.ckpt"), {model._ckpt_name(hp, k): np.array(v) for k, v in tensors.items()})
    hpd = {k: getattr(hp, k) for k in ("imSize", "nClasses", "nChannels", "nExtraConvs", "nLayers", "featMapsFact",
                                       "downSampFact", "ks", "nOut0", "batchSize")}
    hpd["stdDev0"] = 0.03
    pickle.dump(hpd, open(tmp_path / "hp.data", "wb"))
    pickle.dump(0.25, open(tmp_path / "datasetMean.data", "wb"))
    pickle.dump(np.float64(0.125), open(tmp_path / "datasetStDev.data", "wb"))
    art = model.load_model_dir(str(tmp_path), hp.graph)
    assert art.hp == hp and art.mean == 0.25 and art.std == 0.125
    assert np.array_equal(art.blob, blob)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_reads_reference_checkpoints():
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    art = model.load_model_dir(os.path.join(REF, "models", "nucleiDAPI"), model.GRAPH_LEGACY)
    assert art.hp == hp and np.array_equal(art.blob, blob) and (art.mean, art.std) == (mean, std)
    # solo / duo: index + hp only (data shards are downloaded by the reference's Dockerfile) -> loud failure
    with pytest.raises(FileNotFoundError):
        model.load_model_dir(os.path.join(REF, "models", "nucleiDAPI1-5"), model.GRAPH_V2)
    idx = tfckpt.read_index(os.path.join(REF, "models", "nucleiDAPILAMIN", "model.ckpt.index"))
    duo = model.KNOWN_HP["nucleiDAPILAMIN"]
    for name, shape in model.tensor_specs(duo):
        assert idx[model._ckpt_name(duo, name)]["shape"] == tuple(shape), name
    # a v2 graph cannot restore a legacy checkpoint (the reference fails with NotFoundError): KeyError here
    with pytest.raises(KeyError):
        model.blob_from_checkpoint(model.HParams(model.GRAPH_V2, 128, 1, 3, 16, 2, 5, 1),
                                   os.path.join(REF, "models", "nucleiDAPI", "model.ckpt"))


def test_tiff_roundtrip_and_foreign_layouts(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 255, (37, 53), dtype=np.uint8)
    p = str(tmp_path / "t.tif")
    tiffio.imsave(p, a)
    tiffio.imsave(p, a[::-1].copy(), append=True)
    tiffio.imsave(p, (a.astype(np.uint16) * 200), append=True)
    assert tiffio.num_pages(p) == 3
    assert np.array_equal(tiffio.imread(p, 0), a) and np.array_equal(tiffio.imread(p, 1), a[::-1])
    assert np.array_equal(tiffio.imread(p, 2), a.astype(np.uint16) * 200)
    with pytest.raises(IndexError):
        tiffio.imread(p, 3)
    # hand-built big-endian classic TIFF, uint16, two strips (the layout of the reference's sample 105.tif)
    img = rng.integers(0, 65535, (6, 5), dtype=np.uint16)
    be = img.astype(">u2").tobytes()
    strips = [be[:30], be[30:]]
    off0 = 8
    off1 = off0 + len(strips[0])
    ifd_off = off1 + len(strips[1])
    ext = ifd_off + 2 + 9 * 12 + 4

    def ent(tag, typ, cnt, val):
        if typ == 3 and cnt == 1:
            return struct.pack(">HHIHH", tag, typ, cnt, val, 0)
        return struct.pack(">HHII", tag, typ, cnt, val)
    ifd = struct.pack(">H", 9) + ent(256, 3, 1, 5) + ent(257, 3, 1, 6) + ent(258, 3, 1, 16) + ent(259, 3, 1, 1) + \
        ent(273, 4, 2, ext) + ent(277, 3, 1, 1) + ent(278, 3, 1, 3) + ent(279, 4, 2, ext + 8) + ent(339, 3, 1, 1) + \
        struct.pack(">I", 0)
    blob = b"MM" + struct.pack(">HI", 42, ifd_off) + strips[0] + strips[1] + ifd + \
        struct.pack(">II", off0, off1) + struct.pack(">II", 30, 30)
    q = str(tmp_path / "be.tif")
    open(q, "wb").write(blob)
    assert np.array_equal(tiffio.imread(q), img)
    with pytest.raises(NotImplementedError):
        open(str(tmp_path / "x.tif"), "wb").write(b"not a tiff at all")
        tiffio.imread(str(tmp_path / "x.tif"))


def test_band_partition_and_ownership_cover_the_image_once():
    for npr in (1, 2, 3, 7, 11, 86, 342):
        for world in (1, 2, 3, 4, 8):
            bands = sharding.band_partition(npr, world)
            assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == npr
            sizes = [b - a for a, b in bands]
            assert all(x >= 0 for x in sizes) and max(sizes) - min(s for s in sizes if s or world > npr) <= 1 or world > npr
            for (a, b), (c, d) in zip(bands, bands[1:]):
                assert b == c
            for patch in (64, 128, 256):
                m = patch // 8
                sub = patch - 2 * m
                H = npr * sub - 5 if npr * sub > 5 else npr * sub
                if (H + sub - 1) // sub != npr:
                    continue
                rows = [sharding.owned_rows(a, b, npr, sub, m, H) for a, b in bands]
                covered = np.zeros(H, int)
                for y0, y1 in rows:
                    covered[y0:y1] += 1
                assert (covered == 1).all(), (npr, world, patch)
                for (a, b), (y0, y1) in zip(bands, rows):
                    if y1 > y0:   # every tile touching the owned rows lies in patch rows [a-1, b)
                        lo = max(0, (y0 + m - patch + sub) // sub) if y0 + m - patch + 1 > 0 else 0
                        hi = min(npr - 1, (y1 - 1 + m) // sub)
                        assert lo >= max(0, a - 1) and hi <= b - 1


def test_slabs_tile_the_owned_rows_and_only_need_computed_patch_rows():
    """sharding.slab_rows: the slabs of a band partition its owned rows, and slab i only touches patch rows below cut
    i+1 (or the band's last patch row, which is computed first) -- the condition for stitching it before later tiles exist."""
    for npr in (2, 5, 11, 86):
        for world in (1, 2, 3, 8):
            for patch in (32, 256):
                m = patch // 8
                sub = patch - 2 * m
                H = npr * sub - 3
                bands = sharding.band_partition(npr, world)
                active = [b for b in bands if b[1] > b[0]]
                n = max(1, min(4, min(b - a for a, b in active)))
                for a, b in active:
                    y0, y1 = sharding.owned_rows(a, b, npr, sub, m, H)
                    cuts = sharding._slab_cuts(a, b, n)
                    pos = y0
                    for i in range(n):
                        s0, s1 = sharding.slab_rows(a, b, npr, sub, m, H, n, i)
                        assert s0 == pos and s1 >= s0
                        pos = s1
                        if s1 > s0 and i < n - 1:
                            hi = min(npr - 1, (s1 - 1 + m) // sub)     # last patch row touching the slab
                            assert hi < cuts[i + 1]
                    assert pos == y1


def test_hip_runtime_binding_does_not_depend_on_import_order():
    """`from unmicst_amd import umx, sharding; umx.Engine(...)` followed by a later `import torch` must not leave two
    HIP runtimes in the process.  The default binds the runtime the installed PyTorch bundles WITHOUT importing torch (same
    library whichever comes first); `system` is for the per-file tools, and the torch-facing helpers refuse to run on it."""
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch-first":
    import torch
from unmicst_amd import umx, sharding
umx.load()
assert ("torch" in sys.modules) == (order == "torch-first")
import torch
want = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
mode = os.environ.get("UMX_HIP_RUNTIME", "auto")
if mode == "system":
    assert os.path.realpath(umx.hip_runtime) != os.path.realpath(want), umx.hip_runtime
    try:
        umx.require_torch_runtime("sharding.infer_image_sharded")
    except RuntimeError as e:
        assert "ONE HIP runtime" in str(e)
    else:
        raise SystemExit("mixing runtimes was not refused")
else:
    assert os.path.realpath(umx.hip_runtime) == os.path.realpath(want), umx.hip_runtime
    umx.require_torch_runtime("sharding.infer_image_sharded")
print("ok", order, mode)
""" % helpers.ROOT
    for order in ("umx-first", "torch-first"):
        for mode in (None, "system"):
            env = {k: v for k, v in os.environ.items() if k != "UMX_HIP_RUNTIME"}
            if mode:
                env["UMX_HIP_RUNTIME"] = mode
            r = subprocess.run([sys.executable, "-c", code, order], capture_output=True, text=True, timeout=300, env=env)
            assert r.returncode == 0 and "ok " + order in r.stdout, (order, mode, r.stdout[-500:], r.stderr[-1500:])
