#!/usr/bin/env python3
"""CLI dispatcher with the reference's flag surface (reference unmicstWrapper.py:5-22): 1-based --channel /
--classOrder / --GPU are shifted to the scripts' 0-based convention (:35-38) and the tool is chosen by --tool
(:40-61).  The reference re-executes ``python UnMicstX.py ...`` (:90); here the tool runs in-process
(unmicst_amd.driver.run), which is what the reference's own FIXME at :89 asks for."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--tool", help="which UnMicst tool?", default="unmicst-solo")
    p.add_argument("imagePath", help="path to the .tif file")
    p.add_argument("--model", help="type of model. For example, nuclei vs cytoplasm")
    p.add_argument("--outputPath", help="output path of probability map")
    p.add_argument("--channel", help="channel to perform inference on", nargs="+", type=int, default=[1])
    p.add_argument("--classOrder", help="background, contours, foreground", type=int, nargs="+", default=-1)
    p.add_argument("--mean", help="mean intensity of input image. Use -1 to use model", type=float, default=-1)
    p.add_argument("--std", help="mean standard deviation of input image. Use -1 to use model", type=float, default=-1)
    p.add_argument("--scalingFactor", help="factor by which to increase/decrease image size by", type=float, default=1)
    p.add_argument("--stackOutput", help="save probability maps as separate files", action="store_true")
    p.add_argument("--GPU", help="explicitly select GPU", type=int, default=0)
    p.add_argument("--outlier", help="map percentile intensity to max when rescaling intensity values. "
                                     "Max intensity as default", type=float, default=-1)
    p.add_argument("--verbose", help="display error messages for debugging", action="store_true")
    # (not in the reference: A/B timing of its one-pass-per-class loop, unmicst_amd/driver.py)
    p.add_argument("--compat-per-class", dest="compat_per_class", action="store_true",
                   help="A/B timing only: one full inference pass per output class, like the reference")
    return p.parse_args(argv)


def script_argv(a):
    """-> (tool key, argv of the per-tool script): the command line the reference would exec."""
    channel = [c - 1 for c in a.channel]
    tool = a.tool if a.tool in ("unmicst-duo", "unmicst-legacy", "UnMicstCyto2") else "unmicst-solo"
    if tool == "unmicst-legacy":
        print("\nWARNING! YOU HAVE OPTED TO USE UNMICST legacy, WHICH IS GETTING TIRED AND OLD. CONSIDER USING "
              "unmicst-solo OR unmicst-duo (IF YOU ALSO HAVE A NUCLEAR ENVELOPE STAIN\n")
    elif tool == "unmicst-solo":
        print("\nWARNING! USING unmicst-solo AS DEFAULT. THIS MODEL HAS BEEN TRAINED ON MORE TISSUE TYPES. IF YOU WANT "
              "THE LEGACY MODEL, USE --tool unmicst-legacy\n")
    chan = channel[:2] if (tool == "unmicst-duo" and len(channel) == 2) else channel[:1]
    argv = [a.imagePath, "--channel"] + [str(c) for c in chan]
    if a.outputPath is not None:      # the reference forwards the literal string "None"; an absent flag is the intent
        argv += ["--outputPath", str(a.outputPath)]
    argv += ["--mean", str(a.mean), "--std", str(a.std), "--scalingFactor", str(a.scalingFactor),
             "--GPU", str(a.GPU - 1), "--outlier", str(a.outlier)]
    if a.stackOutput:
        argv.append("--stackOutput")
    if a.model:
        argv += ["--model", str(a.model)]
    if a.classOrder != -1:
        argv += ["--classOrder"] + [str(c - 1) for c in a.classOrder[:3]]
    if a.verbose and tool != "UnMicstCyto2":
        argv.append("--verbose")
    if a.compat_per_class:
        argv.append("--compat-per-class")
    return tool, argv


if __name__ == "__main__":
    from unmicst_amd import driver
    tool, argv = script_argv(parse())
    print(driver.TOOLS[tool].script + " " + " ".join(argv))
    sys.exit(driver.run(tool, argv, os.path.dirname(os.path.realpath(__file__))))
