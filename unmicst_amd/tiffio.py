"""Minimal TIFF / BigTIFF / OME-TIFF page reader and BigTIFF writer (numpy only).

Stands in for the third-party readers/writers the reference drivers call at the boundary of the hot path:
``tifffile.imread(path, key=channel)`` / ``skio.imread(path, img_num=channel, plugin='tifffile')``
(reference UnMicst1-5.py:794-797) and ``skimage.io.imsave(path, uint8, bigtiff=True, append=...)``
(reference UnMicst1-5.py:834-862).  A "page" is a top-level IFD (= one channel of an OME-TIFF; pyramid levels
live in SubIFDs and are not top-level pages, matching how tifffile indexes ``key``).

Supported on read: classic TIFF and BigTIFF, both byte orders, strips or tiles, 1 sample per pixel,
uint8/uint16/uint32/int*/float32/float64, compression none (1), LZW (5: the Bio-Formats / OME-TIFF default), deflate
(8 / 32946) or PackBits (32773), with optional horizontal predictor (2).  LZW and PackBits strips are decoded by two host
functions of libumx (no GPU needed).  JPEG / JPEG-2000 / zstd / floating-point predictor raise NotImplementedError (the
reference raises NotImplementedError for file types it cannot read, UnMicst1-5.py:785-806).
"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

_TYPE_FMT = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d",
             16: "Q", 17: "q", 18: "Q"}


class _Reader:
    def __init__(self, path: str):
        self.f = open(path, "rb")
        head = self.f.read(16)
        if head[:2] == b"II":
            self.bo = "<"
        elif head[:2] == b"MM":
            self.bo = ">"
        else:
            raise NotImplementedError("not a TIFF file: %s" % path)
        (magic,) = struct.unpack(self.bo + "H", head[2:4])
        if magic == 42:
            self.big = False
            (self.first_ifd,) = struct.unpack(self.bo + "I", head[4:8])
        elif magic == 43:
            self.big = True
            (self.first_ifd,) = struct.unpack(self.bo + "Q", head[8:16])
        else:
            raise NotImplementedError("bad TIFF magic %d" % magic)

    def close(self):
        self.f.close()

    def _read_ifd(self, offset: int) -> Tuple[Dict[int, tuple], int]:
        f, bo = self.f, self.bo
        f.seek(offset)
        if self.big:
            (n,) = struct.unpack(bo + "Q", f.read(8))
            esz, cfmt, inline = 20, "Q", 8
        else:
            (n,) = struct.unpack(bo + "H", f.read(2))
            esz, cfmt, inline = 12, "I", 4
        raw = f.read(n * esz)
        nxt = struct.unpack(bo + cfmt, f.read(inline))[0]
        tags: Dict[int, tuple] = {}
        for i in range(n):
            e = raw[i * esz:(i + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            (count,) = struct.unpack(bo + cfmt, e[4:4 + inline])
            fmt = _TYPE_FMT.get(typ)
            if fmt is None:
                continue
            nper = len(fmt)
            size = struct.calcsize("=" + fmt) * count
            if size <= inline:
                data = e[4 + inline:4 + inline + size]
            else:
                (off,) = struct.unpack(bo + cfmt, e[4 + inline:4 + 2 * inline])
                pos = f.tell()
                f.seek(off)
                data = f.read(size)
                f.seek(pos)
            if typ == 2:
                tags[tag] = (data.rstrip(b"\0"),)
            else:
                tags[tag] = struct.unpack(bo + fmt[0] * (count * nper), data)
        return tags, nxt

    def ifds(self) -> List[Dict[int, tuple]]:
        out = []
        off = self.first_ifd
        seen = set()
        while off and off not in seen:
            seen.add(off)
            tags, off = self._read_ifd(off)
            out.append(tags)
        return out

    def page(self, tags: Dict[int, tuple]) -> np.ndarray:
        def one(tag, default=None):
            v = tags.get(tag)
            return v[0] if v else default

        width, height = one(256), one(257)
        bits = one(258, 1)
        comp = one(259, 1)
        spp = one(277, 1)
        fmt = one(339, 1)
        predictor = one(317, 1)
        if spp != 1:
            raise NotImplementedError("only single-sample (grayscale) pages are supported")
        kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
        if kind is None or bits % 8:
            raise NotImplementedError("unsupported sample format %s / %s bits" % (fmt, bits))
        dt = np.dtype(self.bo + kind + str(bits // 8))
        if comp not in (1, 5, 8, 32946, 32773):
            raise NotImplementedError("unsupported TIFF compression %d" % comp)

        def decode(buf: bytes, rows: int, cols: int) -> np.ndarray:
            if comp in (8, 32946):
                buf = zlib.decompress(buf)
            elif comp in (5, 32773):
                from . import umx
                buf = umx.tiff_decode("lzw" if comp == 5 else "packbits", buf, rows * cols * dt.itemsize)
            a = np.frombuffer(buf, dtype=dt, count=rows * cols).reshape(rows, cols)
            if predictor == 2:
                a = np.cumsum(a, axis=1, dtype=a.dtype.newbyteorder("="))
            elif predictor != 1:
                raise NotImplementedError("unsupported predictor %d" % predictor)
            return a

        out = np.empty((height, width), dtype=dt.newbyteorder("="))
        f = self.f
        if 322 in tags:  # tiled
            tw, th = one(322), one(323)
            offs, cnts = tags[324], tags[325]
            tiles_x = (width + tw - 1) // tw
            for i, (o, c) in enumerate(zip(offs, cnts)):
                ty, tx = divmod(i, tiles_x)
                f.seek(o)
                t = decode(f.read(c), th, tw)
                r0, c0 = ty * th, tx * tw
                out[r0:r0 + th, c0:c0 + tw] = t[:max(0, min(th, height - r0)), :max(0, min(tw, width - c0))]
        else:
            rps = min(one(278, height), height)
            offs, cnts = tags[273], tags[279]
            for i, (o, c) in enumerate(zip(offs, cnts)):
                r0 = i * rps
                rows = min(rps, height - r0)
                f.seek(o)
                out[r0:r0 + rows] = decode(f.read(c), rows, width)
        return out


def num_pages(path: str) -> int:
    r = _Reader(path)
    try:
        return len(r.ifds())
    finally:
        r.close()


def imread(path: str, key: int = 0) -> np.ndarray:
    """Read top-level page ``key`` (== tifffile.imread(path, key=key))."""
    r = _Reader(path)
    try:
        ifds = r.ifds()
        if key < 0 or key >= len(ifds):
            raise IndexError("page %d out of range (file has %d pages)" % (key, len(ifds)))
        return r.page(ifds[key])
    finally:
        r.close()


def imread_all(path: str) -> np.ndarray:
    r = _Reader(path)
    try:
        return np.stack([r.page(t) for t in r.ifds()])
    finally:
        r.close()


def _ifd_bytes(img: np.ndarray, data_offset: int) -> bytes:
    """One BigTIFF IFD (little-endian) for an uncompressed single-strip grayscale page; next-IFD offset = 0."""
    h, w = img.shape
    kind = {"u": 1, "i": 2, "f": 3}[img.dtype.kind]
    entries = [
        (256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, img.dtype.itemsize * 8), (259, 3, 1, 1), (262, 3, 1, 1),
        (273, 16, 1, data_offset), (277, 3, 1, 1), (278, 4, 1, h), (279, 16, 1, img.nbytes), (339, 3, 1, kind),
    ]
    out = struct.pack("<Q", len(entries))
    for tag, typ, cnt, val in entries:
        out += struct.pack("<HHQQ", tag, typ, cnt, val)
    out += struct.pack("<Q", 0)
    return out


def imsave(path: str, img: np.ndarray, append: bool = False) -> None:
    """Write (append=False) or append one 2-D page to a little-endian BigTIFF.

    Same call pattern as the reference's ``skimage.io.imsave(path, page, bigtiff=True, append=..)``.
    """
    img = np.ascontiguousarray(img)
    if img.ndim != 2:
        raise ValueError("imsave writes one 2-D page per call")
    img = img.astype(img.dtype.newbyteorder("<"), copy=False)
    if append and os.path.exists(path):
        with open(path, "r+b") as f:
            head = f.read(16)
            if head[:4] != b"II\x2b\x00":
                raise NotImplementedError("can only append to little-endian BigTIFF files written by this module")
            # walk to the last IFD's next-pointer
            ptr_pos = 8
            (off,) = struct.unpack("<Q", head[8:16])
            while off:
                f.seek(off)
                (n,) = struct.unpack("<Q", f.read(8))
                ptr_pos = off + 8 + 20 * n
                f.seek(ptr_pos)
                (off,) = struct.unpack("<Q", f.read(8))
            f.seek(0, 2)
            pos = f.tell()
            pos += (-pos) % 16
            f.seek(pos)
            f.write(memoryview(img).cast("B"))       # (no tobytes(): a page of a 16384 x 16384 slide is 268 MB)
            ifd_pos = f.tell()
            ifd_pos += (-ifd_pos) % 8
            f.seek(ifd_pos)
            f.write(_ifd_bytes(img, pos))
            f.seek(ptr_pos)
            f.write(struct.pack("<Q", ifd_pos))
        return
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HHHQ", 43, 8, 0, 0))
        pos = 16
        f.write(memoryview(img).cast("B"))
        ifd_pos = pos + img.nbytes
        ifd_pos += (-ifd_pos) % 8
        f.seek(ifd_pos)
        f.write(_ifd_bytes(img, pos))
        f.seek(8)
        f.write(struct.pack("<Q", ifd_pos))
