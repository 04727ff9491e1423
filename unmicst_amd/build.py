"""Build libumx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No other target is supported."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libumx.so")
SOURCES = ["umx_kernels.hip", "umx_conv_f16.hip", "umx_engine.hip", "umx_train_kernels.hip", "umx_train.hip"]
DEPS = SOURCES + ["umx_kernels.h", os.path.join("..", "..", "include", "umx.h"),
                  os.path.join("..", "..", "include", "umx_train.h")]


def lib_path() -> str:
    return LIB


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if os.path.getmtime(os.path.abspath(__file__)) > t:
        return True
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden",
           # no DT_NEEDED on a HIP runtime: the loader (umx.load) binds libumx to the ONE libamdhip64 of the process
           # (PyTorch bundles its own copy; two HIP/HSA runtimes in one process cannot both drive the GPU)
           "-no-hip-rt",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
