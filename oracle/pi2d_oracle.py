"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by unmicst_amd/).

numpy restatement of the reference's tile partitioner / stitcher and of the whole-image inference
loop, written from the reference's semantics:

* ``PI2D`` state machine          reference toolbox/PartitionOfImage.py:23-122
* ``singleImageInference`` loop   reference UnMicst1-5.py:687-710 (solo), UnMicst2.py:666-689 (duo),
                                   UnMicst.py:520-541 (legacy)

Pinning: tests/golden/pi2d_*.npz hold outputs of the *imported reference PI2D* (generated in the build
container by tools/make_golden.py, which stubs the absent tifffile/skimage imports); tests/test_oracle_golden.py
checks this restatement against them bit-for-bit, and against the reference's "UNet sample data" goldens end to end.

Everything numeric is kept in the reference's dtypes: float64 image / blend window, float16 accumulators
updated with numpy's ``float16 += float64`` semantics (compute in float64, round to float16 per tile).
"""
from __future__ import annotations

import numpy as np


class PI2DOracle:
    """Instance-based restatement of the reference's static ``PI2D`` class."""

    def __init__(self, image: np.ndarray, patch_size: int, margin: int, mode: str):
        # reference PartitionOfImage.py:23-75
        self.image = image
        self.patch = patch_size
        self.margin = margin
        sub = patch_size - 2 * margin
        self.sub = sub
        self.W = self.blend_window(patch_size, margin)
        if image.ndim == 2:
            nr, nc = image.shape
        elif image.ndim == 3:
            nz, nr, nc = image.shape
        else:
            raise ValueError("image must be 2-D or channel-first 3-D")
        self.nr, self.nc = nr, nc
        self.npr = int(np.ceil(nr / sub))
        self.npc = int(np.ceil(nc / sub))
        self.nrpi = self.npr * sub + 2 * margin
        self.ncpi = self.npc * sub + 2 * margin
        if image.ndim == 2:
            self.padded = np.zeros((self.nrpi, self.ncpi))
            self.padded[margin:margin + nr, margin:margin + nc] = image
        else:
            self.padded = np.zeros((nz, self.nrpi, self.ncpi))
            self.padded[:, margin:margin + nr, margin:margin + nc] = image
        self.pc = []
        for i in range(self.npr):
            r0 = i * sub
            for j in range(self.npc):
                c0 = j * sub
                self.pc.append([r0, r0 + patch_size, c0, c0 + patch_size])
        self.num_patches = len(self.pc)
        self.mode = mode
        self.output = None
        self.count = None

    @staticmethod
    def blend_window(patch_size: int, margin: int) -> np.ndarray:
        # reference PartitionOfImage.py:30-39: concentric rings, ring i has weight i/(2*margin), outermost ring 0
        W = np.ones((patch_size, patch_size))
        W[[0, -1], :] = 0
        W[:, [0, -1]] = 0
        for i in range(1, 2 * margin):
            v = i / (2 * margin)
            W[i, i:-i] = v
            W[-i - 1, i:-i] = v
            W[i:-i, i] = v
            W[i:-i, -i - 1] = v
        return W

    def get_patch(self, i: int) -> np.ndarray:
        r0, r1, c0, c1 = self.pc[i]
        if self.padded.ndim == 2:
            return self.padded[r0:r1, c0:c1]
        return self.padded[:, r0:r1, c0:c1]

    def create_output(self, n_channels: int = 1) -> None:
        # reference PartitionOfImage.py:84-90 (only the single-plane form is used by singleImageInference)
        if n_channels == 1:
            self.output = np.zeros((self.nrpi, self.ncpi), np.float16)
        else:
            self.output = np.zeros((n_channels, self.nrpi, self.ncpi), np.float16)
        if self.mode == "accumulate":
            self.count = np.zeros((self.nrpi, self.ncpi), np.float16)

    def patch_output(self, i: int, P: np.ndarray) -> None:
        # reference PartitionOfImage.py:92-106
        r0, r1, c0, c1 = self.pc[i]
        if self.mode == "accumulate":
            self.count[r0:r1, c0:c1] += self.W
        if P.ndim == 2:
            if self.mode == "accumulate":
                self.output[r0:r1, c0:c1] += np.multiply(P, self.W)
            elif self.mode == "replace":
                self.output[r0:r1, c0:c1] = P
        else:
            if self.mode == "accumulate":
                for k in range(P.shape[0]):
                    self.output[k, r0:r1, c0:c1] += np.multiply(P[k], self.W)
            elif self.mode == "replace":
                self.output[:, r0:r1, c0:c1] = P

    def get_valid_output(self) -> np.ndarray:
        # reference PartitionOfImage.py:108-122
        m, nr, nc = self.margin, self.nr, self.nc
        if self.output.ndim == 2:
            if self.mode == "accumulate":
                with np.errstate(divide="ignore", invalid="ignore"):
                    return np.divide(self.output[m:m + nr, m:m + nc], self.count[m:m + nr, m:m + nc])
            return self.output[m:m + nr, m:m + nc]
        if self.mode == "accumulate":
            C = self.count[m:m + nr, m:m + nc]
            for k in range(self.output.shape[0]):
                with np.errstate(divide="ignore", invalid="ignore"):
                    self.output[k, m:m + nr, m:m + nc] = np.divide(self.output[k, m:m + nr, m:m + nc], C)
        return self.output[:, m:m + nr, m:m + nc]


def normalised_batch(pi: PI2DOracle, first: int, count: int, n_channels: int, mean: float, std: float,
                     duplicate_plane: bool) -> np.ndarray:
    """Tiles [first, first+count) as the float32 NHWC batch TensorFlow is fed.

    reference UnMicst1-5.py:700-702 (solo: the 2-D patch is copied to every channel), UnMicst2.py:679-681
    (duo: P[iChan]), UnMicst.py:533 (legacy, one channel): ``(patch - mean)/std`` in float64; TF casts the
    float64 feed to the float32 placeholder.
    """
    P = pi.patch
    batch = np.zeros((count, P, P, n_channels))
    for k in range(count):
        p = (pi.get_patch(first + k) - mean) / std
        for c in range(n_channels):
            batch[k, :, :, c] = p if (duplicate_plane or p.ndim == 2) else p[c]
    return batch.astype(np.float32)


def single_image_inference(image: np.ndarray, forward, patch: int, n_channels: int, mean: float, std: float,
                           mode: str, pm_index: int, batch_size: int, duplicate_plane: bool = False) -> np.ndarray:
    """The reference's hot loop: returns the float16 plane of class ``pm_index``.

    ``forward(batch_f32_nhwc) -> probs_f32 [B,P,P,K]`` stands for ``Session.run(UNet2D.nn, ...)``.
    Batching differs from the reference only in that the last, partial batch is fed at its true size (the
    reference feeds stale tiles in the unused slots; convolutions are per-sample so used outputs are identical).
    """
    pi = PI2DOracle(image, patch, int(patch / 8), mode)
    pi.create_output(1)
    i = 0
    while i < pi.num_patches:
        n = min(batch_size, pi.num_patches - i)
        out = forward(normalised_batch(pi, i, n, n_channels, mean, std, duplicate_plane))
        for k in range(n):
            pi.patch_output(i + k, out[k, :, :, pm_index])
        i += n
    return pi.get_valid_output()


def stitch_all_classes(image_shape, patch: int, probs: np.ndarray, mode: str = "accumulate") -> np.ndarray:
    """Stitch per-tile probabilities [T,P,P,K] into [K,H,W] float16 planes, one reference pass per class."""
    H, W = image_shape
    dummy = np.zeros((H, W))
    planes = []
    for k in range(probs.shape[-1]):
        pi = PI2DOracle(dummy, patch, int(patch / 8), mode)
        pi.create_output(1)
        for t in range(pi.num_patches):
            pi.patch_output(t, probs[t, :, :, k])
        planes.append(np.array(pi.get_valid_output()))
    return np.stack(planes)
