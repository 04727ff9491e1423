"""One command-line driver for every UnMicst tool.

The reference ships the same ``__main__`` block five times with small per-tool differences (UnMicst1-5.py:713-876 solo,
UnMicst2.py:692-835 duo, UnMicst.py:544-678 legacy, UnMicstCyto2.py:679-827).  Here the differences are data
(``TOOLS``) and the flow is written once: read page(s) -> resize -> intensity rescale -> ONE pass of the HIP engine for
all classes -> uint8 recipe -> BigTIFF pages with the reference's names and page order.

Kept identical on purpose (file-level parity): argparse surfaces; 0-based channels in the scripts / 1-based in the
wrapper; the file-type switch on the text after the first dot (solo: after the last dots, with the ``ome.`` special
case); float32 input cast to uint16; the solo quirk that the network sees the *un-rescaled* image (``cells = I`` is
bound before the rescale, UnMicst1-5.py:816-821) while legacy/duo/cyto feed the rescaled one; uint8 truncation twice
(UnMicst1-5.py:848-854); ``_Probabilities_`` pages in reversed class order; ``qc/<stem>_Preview_`` = [contours, raw/max];
default output directory ``<parent of parent>/probability_maps``.

Not kept: ``.czi`` / ``.nd2`` (proprietary readers absent: NotImplementedError, which is what the reference raises for
types it cannot read, UnMicst1-5.py:806); the NVML device probe (replaced by umx_device_mem_info).
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from dataclasses import dataclass

import numpy as np

from . import imtools, tiffio
from .unet2d import UNet2D


@dataclass(frozen=True)
class ToolSpec:
    script: str              # reference script this spec mirrors
    default_model: str
    graph: object            # model.GRAPH_* or None (detect)
    multi_channel: bool      # --channel nargs='+' (solo, duo) vs a single int (legacy, cyto)
    n_inputs: int            # image planes fed to the network (duo: 2)
    split_last: bool         # solo derives stem/type from the LAST dots; the others split at the first dot
    tiff_types: tuple        # extensions read as OME/BigTIFF pages
    infer_rescaled: bool     # False: solo's `cells = I` quirk
    cast_float32: bool
    has_verbose: bool
    qc_dir: bool             # Preview goes to <out>/qc
    suffix_plus_one: bool    # file-name suffix is channel+1 (cyto writes the raw 0-based channel)


TOOLS = {
    "unmicst-solo": ToolSpec("UnMicst1-5.py", "nucleiDAPI1-5", 1, True, 1, True, ("ome.tif", "ome.tiff", "btf"), False,
                             True, True, True, True),
    "unmicst-duo": ToolSpec("UnMicst2.py", "nucleiDAPILAMIN", 1, True, 2, False, ("ome.tif", "ome.tiff", "btf"), True,
                            True, True, True, True),
    "unmicst-legacy": ToolSpec("UnMicst.py", "nucleiDAPI", 0, False, 1, False, ("ome.tif", "ome.tiff", "btf"), True,
                               True, True, True, True),
    "UnMicstCyto2": ToolSpec("UnMicstCyto2.py", "nucleiDAPI", 1, False, 1, False, ("ome.tif", "btf"), True, False,
                             False, False, False),
}


def build_parser(spec: ToolSpec) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog=spec.script)
    p.add_argument("imagePath", help="path to the .tif file")
    p.add_argument("--model", help="type of model. For example, nuclei vs cytoplasm", default=spec.default_model)
    p.add_argument("--outputPath", help="output path of probability map")
    if spec.multi_channel:
        p.add_argument("--channel", help="channel to perform inference on", nargs="+", default=[0])
    else:
        p.add_argument("--channel", help="channel to perform inference on", type=int, default=0)
    p.add_argument("--classOrder", help="background, contours, foreground", type=int, nargs="+", default=-1)
    p.add_argument("--mean", help="mean intensity of input image. Use -1 to use model", type=float, default=-1)
    p.add_argument("--std", help="mean standard deviation of input image. Use -1 to use model", type=float, default=-1)
    p.add_argument("--scalingFactor", help="factor by which to increase/decrease image size by", type=float, default=1)
    p.add_argument("--stackOutput", help="save probability maps as separate files", action="store_true")
    p.add_argument("--GPU", help="explicitly select GPU", type=int, default=-1)
    p.add_argument("--outlier", help="map percentile intensity to max when rescaling intensity values. "
                                     "Max intensity as default", type=float, default=-1)
    if spec.has_verbose:
        p.add_argument("--verbose", help="display error messages for debugging", action="store_true")
    # the one flag the reference does not have: the reference runs the whole slide once PER CLASS
    # (UnMicst1-5.py:845-863); here one pass yields every class.  For A/B timing this switch takes the reference's shape:
    # host-side pre-processing, then one full pass of the engine per class written.  Same bytes in the files.
    p.add_argument("--compat-per-class", dest="compat_per_class", action="store_true",
                   help="A/B timing only: one full inference pass per output class, like the reference (same output bytes)")
    return p


def models_root(script_dir: str) -> str:
    """``<script dir>/models`` like the reference (UnMicst1-5.py:744), overridable with UMX_MODELS_DIR."""
    return os.environ.get("UMX_MODELS_DIR") or os.path.join(script_dir, "models")


def split_name(file_name: str, spec: ToolSpec):
    """-> (stem, type) the way the tool's reference script derives them."""
    if spec.split_last:
        parts = file_name.split(os.extsep)
        if len(parts) < 2:
            raise NotImplementedError("Input filename has no extension")
        if parts[-2] == "ome":
            return os.extsep.join(parts[:-2]), os.extsep.join(parts[-2:])
        return os.extsep.join(parts[:-1]), parts[-1]
    parts = file_name.split(os.extsep, 1)
    if len(parts) < 2:
        raise NotImplementedError("Input filename has no extension")
    return parts[0], parts[1]


def read_plane(path: str, file_type: str, channel: int, spec: ToolSpec) -> np.ndarray:
    if file_type in spec.tiff_types or file_type == "tif":
        I = tiffio.imread(path, key=int(channel))
    else:
        raise NotImplementedError("Don't know how to read image with extension .%s" % file_type)
    if spec.cast_float32 and I.dtype == np.float32:
        I = np.uint16(I)
    return I


def preprocess(I: np.ndarray, scaling: float, outlier: float):
    """resize by --scalingFactor, then rescale intensities to (0, 0.983) -> (resized float image, rescaled image)."""
    hsize = int(float(I.shape[0]) * float(scaling))
    vsize = int(float(I.shape[1]) * float(scaling))
    R = imtools.resize(I, (hsize, vsize))
    max_limit = np.max(R) if outlier == -1 else np.percentile(R, outlier)
    S = imtools.im2double(imtools.rescale_intensity(R, (np.min(R), max_limit), (0, 0.983)))
    return R, S


class _Stages:
    """UMX_CLI_TIMING=1: one JSON line on stderr with the seconds each stage of the run took (tools/cli_walltime.py reads it)."""

    def __init__(self):
        self.on = bool(os.environ.get("UMX_CLI_TIMING"))
        self.t = time.perf_counter()
        self.stages = []
        t0 = float(os.environ.get("UMX_CLI_T0", "0") or 0)
        if self.on and t0:
            self.stages.append(("interpreter_start_and_imports", time.time() - t0))

    def mark(self, name):
        if self.on:
            now = time.perf_counter()
            self.stages.append((name, now - self.t))
            self.t = now

    def report(self):
        if self.on:
            import json
            import resource
            t0 = float(os.environ.get("UMX_CLI_T0", "0") or 0)
            d = {k: round(v, 4) for k, v in self.stages}
            if t0:
                d["since_parent_spawned"] = round(time.time() - t0, 4)
            d["user_cpu_s"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_utime, 3)
            print("umx-cli-timing " + json.dumps(d), file=sys.stderr)


def plane_range(raw: np.ndarray):
    """(min, max) of a plane in one pass on host threads (umx_plane_range): the reference's np.min / np.max over the page
    (UnMicst1-5.py:817-821), 2 x 70 ms in numpy for a 16384 x 16384 uint16 page.  The fast path hands the pair to the engine
    (umx_infer_image_raw_range: the upload then overlaps the inference) and reuses the maximum for the preview page."""
    a = np.asarray(raw)
    if a.dtype in (np.uint8, np.uint16) and a.size >= (1 << 20):
        from . import umx as _umx
        return _umx.plane_range(a)
    return int(a.min()), int(a.max())


def preview_u8(raw: np.ndarray, top_value=None) -> np.ndarray:
    """np.uint8(255 * (im2double(raw) / max)) (reference UnMicst1-5.py:809-810,861): a function of the raw value alone, so it is
    evaluated once per possible value with the same float64 operations and looked up -- four float64 passes over a 16384 x
    16384 plane (2.1 GB each) were 1.5 s of the tool's 3 s."""
    if raw.dtype in (np.uint8, np.uint16) and raw.size > (1 << 16):
        top = imtools.im2double(np.asarray(raw.max() if top_value is None else top_value, dtype=raw.dtype))
        vals = imtools.im2double(np.arange(np.iinfo(raw.dtype).max + 1, dtype=raw.dtype))
        return np.uint8(255 * (vals / top))[raw]
    rawI = imtools.im2double(raw)
    return np.uint8(255 * (rawI / np.max(rawI)))


def run(tool: str, argv=None, script_dir: str = None) -> int:
    spec = TOOLS[tool]
    st = _Stages()
    args = build_parser(spec).parse_args(argv)
    script_dir = script_dir or os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    model_path = args.model if os.path.isdir(args.model) else os.path.join(models_root(script_dir), args.model)

    def channel_list():
        ch = [int(c) for c in args.channel] if spec.multi_channel else [int(args.channel)]
        return ([ch[0], ch[0]] if len(ch) == 1 else ch[:2]) if spec.n_inputs == 2 else [ch[0]]

    # the page(s) are read on a thread while the engine is set up (model load, planning, weight packing: 0.2 - 0.6 s against
    # 0.18 s of file reading for a 16384 x 16384 page; both release the GIL).  An error of the read surfaces where the reference
    # reads, after the set-up.
    from concurrent.futures import ThreadPoolExecutor
    _pool = ThreadPoolExecutor(1)
    _ftype = split_name(os.path.basename(args.imagePath), spec)[1]
    _pages = _pool.submit(lambda: [read_plane(args.imagePath, _ftype, ch, spec) for ch in channel_list()])
    if args.GPU == -1:
        print("automatically choosing GPU")
    try:
        UNet2D.singleImageInferenceSetup(model_path, args.GPU, args.mean, args.std, graph=None)
    except BaseException:
        _pool.shutdown(wait=True)
        raise
    print("Using GPU " + str(UNet2D.Engine.device))
    st.mark("setup")
    reuse_before = UNet2D.reuse_pass
    try:
        n_class = UNet2D.hp["nClasses"]
        image_path = args.imagePath
        channels = [int(c) for c in args.channel] if spec.multi_channel else [int(args.channel)]
        first = channels[0]
        if spec.n_inputs == 2:
            channels = [first, first] if len(channels) == 1 else channels[:2]
            print("Using channels %d and %d" % (channels[0] + 1, channels[1] + 1))
        else:
            channels = [first]
            if spec.multi_channel:
                print("Using channel " + str(first + 1))
        parent = os.path.dirname(os.path.dirname(image_path))
        stem, file_type = split_name(os.path.basename(image_path), spec)

        top_value = None                      # max of the preview plane where the fast path has found it already
        try:
            raws = _pages.result()
        finally:
            _pool.shutdown(wait=True)
        raw = raws[-1]                        # the reference keeps the last plane read for the preview (rawI)
        raw_shape = raw.shape[:2]
        class_order = range(n_class) if args.classOrder == -1 else args.classOrder
        st.mark("read")

        out_dir = args.outputPath if args.outputPath else parent + "//probability_maps"
        os.makedirs(out_dir, exist_ok=True)
        qc_dir = out_dir + "//qc" if spec.qc_dir else out_dir
        if spec.qc_dir:
            os.makedirs(qc_dir, exist_ok=True)
        suffix = str(first + 1) if spec.suffix_plus_one else str(first)

        # fast path: the whole pre/post-processing -- im2double, both resizes at --scalingFactor != 1, min/max (or the
        # --outlier percentile, exact by radix selection) + rescale, the double uint8 cast -- runs on the GPU next to the
        # inference (umx_infer_image_raw / _raw_scaled / _raw_outlier): no float64 upload / float16 download.
        fast = (all(r.dtype in (np.uint8, np.uint16) for r in raws)
                and len({(r.shape, r.dtype) for r in raws}) == 1 and not os.environ.get("UMX_NO_RAW_PATH")
                and not args.compat_per_class)
        if args.compat_per_class:
            UNet2D.reuse_pass = False            # every singleImageInference below is a whole pass (UnMicst1-5.py:848,867,872)
        preview_job = None
        if fast:
            stack_raw = np.stack(raws) if spec.n_inputs == 2 else raws[0]
            # the preview page (a table look-up over the raw plane, 0.1 s for 16384 x 16384) on a thread under the engine call
            if raw.size > (1 << 22):
                _pool2 = ThreadPoolExecutor(1)
                preview_job = _pool2.submit(lambda: preview_u8(raw, plane_range(raw)[1]))
                _pool2.shutdown(wait=False)
            if args.outlier != -1 and spec.infer_rescaled:   # (a tool that feeds the un-rescaled plane ignores the limit)
                u8_planes = UNet2D.singleImageInferenceRawOutlier(stack_raw, args.scalingFactor, args.outlier, "accumulate")
            elif float(args.scalingFactor) == 1.0:
                # (the page's extrema from the host pass the reference makes too: the engine then starts on the first rows while
                # the rest of the page is still crossing the bus)
                ranges = [plane_range(r) for r in raws] if spec.infer_rescaled else None
                if ranges:
                    top_value = ranges[-1][1]
                u8_planes = UNet2D.singleImageInferenceRaw(stack_raw, spec.infer_rescaled, "accumulate", value_range=ranges)
            else:   # both resizes (skimage defaults) run on the device too
                u8_planes = UNet2D.singleImageInferenceRawScaled(stack_raw, args.scalingFactor, spec.infer_rescaled, "accumulate")

            def plane_u8(k):
                return u8_planes[k]
            st.mark("engine")
        else:
            planes_in = []
            for r in raws:
                resized, rescaled = preprocess(r, args.scalingFactor, args.outlier)
                planes_in.append(rescaled if spec.infer_rescaled else resized)
            cells = np.stack(planes_in) if spec.n_inputs == 2 else planes_in[0]

            def plane_u8(k):
                return imtools.to_uint8_via_resize(UNet2D.singleImageInference(cells, "accumulate", k), raw_shape)

        if args.stackOutput:
            stack = out_dir + "//" + stem + "_Probabilities_" + suffix + ".tif"
            preview = qc_dir + "//" + stem + "_Preview_" + suffix + ".tif"
            for page, k in enumerate(class_order[::-1]):   # reversed "to align with ilastik" (UnMicst1-5.py:848)
                pm = plane_u8(k)
                tiffio.imsave(stack, pm, append=page > 0)
                if page == 1:
                    tiffio.imsave(preview, pm, append=False)
                    tiffio.imsave(preview, preview_job.result() if preview_job else preview_u8(raw, top_value), append=True)
        else:
            cont = out_dir + "//" + stem + "_ContoursPM_" + suffix + ".tif"
            tiffio.imsave(cont, plane_u8(class_order[1]), append=False)
            tiffio.imsave(cont, preview_job.result() if preview_job else preview_u8(raw, top_value), append=True)
            tiffio.imsave(out_dir + "//" + stem + "_NucleiPM_" + suffix + ".tif", plane_u8(class_order[2]), append=False)
        st.mark("write")
    finally:
        if args.compat_per_class:
            UNet2D.reuse_pass = reuse_before
        UNet2D.singleImageInferenceCleanup()
        st.mark("cleanup")
        st.report()
    return 0


def main(tool: str, script_file: str) -> None:
    # the per-file tools never import torch: they take the system ROCm runtime (umx._bind_hip_runtime: `system`; in every other
    # process libumx binds the runtime an installed PyTorch bundles, so that a later `import torch` shares it)
    os.environ.setdefault("UMX_HIP_RUNTIME", "system")
    rc = run(tool, None, os.path.dirname(os.path.realpath(script_file)))
    # every output file is written and closed, the engine destroyed: what is left is interpreter finalisation and the unloading of the
    # HIP runtime, 0.2 s of a 0.9 s tool.  UMX_NO_FAST_EXIT=1 leaves through sys.exit as usual.
    if os.environ.get("UMX_NO_FAST_EXIT"):
        sys.exit(rc)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(rc)
