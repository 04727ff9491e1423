"""The drivers' TIFF reader against files written by an independent implementation (Pillow / libtiff): LZW (the
Bio-Formats / OME-TIFF default), PackBits and deflate strips, uint8 / uint16 -- the reference reads these through tifffile
(UnMicst1-5.py:794-797).  The LZW and PackBits decoders are host functions of libumx (no GPU needed)."""
import numpy as np
import pytest

from unmicst_amd import tiffio, umx

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def _cases():
    rng = np.random.default_rng(3)
    for dt in (np.uint8, np.uint16):
        top = np.iinfo(dt).max
        yield "ramp-%s" % dt.__name__, (np.arange(301 * 203) % top).astype(dt).reshape(301, 203)
        yield "noise-%s" % dt.__name__, rng.integers(0, top, (257, 129)).astype(dt)     # incompressible: table resets
        yield "flat-%s" % dt.__name__, np.full((64, 70), 7, dt)                          # long runs: KwKwK codes
        blobs = np.zeros((400, 333), dt)
        blobs[50:200, 40:300] = rng.integers(0, 9, (150, 260)).astype(dt) * (top // 16)
        yield "blobs-%s" % dt.__name__, blobs


@pytest.mark.parametrize("comp", ["tiff_lzw", "packbits", "tiff_adobe_deflate", "raw"])
def test_reads_what_libtiff_writes(comp, tmp_path):
    for name, a in _cases():
        path = str(tmp_path / ("%s_%s.tif" % (name, comp)))
        Image.fromarray(a).save(path, compression=comp)
        b = tiffio.imread(path)
        assert b.dtype == a.dtype and b.shape == a.shape, (name, comp)
        assert np.array_equal(a, b), (name, comp)


def test_multi_page_lzw_stack(tmp_path):
    rng = np.random.default_rng(5)
    pages = [rng.integers(0, 60000, (90, 120)).astype(np.uint16) for _ in range(3)]
    path = str(tmp_path / "stack.ome.tif")
    Image.fromarray(pages[0]).save(path, compression="tiff_lzw", save_all=True,
                                   append_images=[Image.fromarray(p) for p in pages[1:]])
    assert tiffio.num_pages(path) == 3
    for k, p in enumerate(pages):
        assert np.array_equal(tiffio.imread(path, key=k), p)


def test_decoders_reject_garbage_and_respect_capacity():
    assert umx.tiff_decode("packbits", bytes([2, 1, 2, 3, 0xFE, 9]), 6) == bytes([1, 2, 3, 9, 9, 9])
    with pytest.raises(ValueError):
        umx.tiff_decode("packbits", bytes([5, 1]), 16)                   # literal run longer than the stream
    with pytest.raises(ValueError):
        umx.tiff_decode("lzw", bytes([0xFF, 0xFF, 0xFF, 0xFF]), 16)      # code beyond the table
    # 9-bit codes MSB first: Clear(256) 'A'(65) 'B'(66) 258("AB") EOI(257)
    bits = "".join(format(c, "09b") for c in (256, 65, 66, 258, 257))
    bits += "0" * (-len(bits) % 8)
    stream = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    assert umx.tiff_decode("lzw", stream, 4) == b"ABAB"
