"""Build libumx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No other target is supported.

Every source is compiled to its own object (unmicst_amd/_obj/, git-ignored) -- in parallel, and only when it or a header
changed -- and the objects are linked into unmicst_amd/libumx.so."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libumx.so")
SOURCES = ["umx_kernels.hip", "umx_conv_f16.hip", "umx_conv_first.hip", "umx_graph.hip", "umx_plan.hip", "umx_engine.hip", "umx_host.hip",
           "umx_tiff.hip", "umx_shard.hip", "umx_train_kernels.hip", "umx_train.hip"]
HEADERS = ["umx_kernels.h", "umx_internal.h", os.path.join("..", "..", "include", "umx.h"), os.path.join("..", "..", "include", "umx_train.h")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Wno-unused-value", "-Wno-unused-result"]


def lib_path() -> str:
    return LIB


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    deps = [os.path.abspath(__file__)] + [os.path.join(CSRC, d) for d in SOURCES + HEADERS]
    return _stale(LIB, deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # no DT_NEEDED on a HIP runtime: the loader (umx.load) binds libumx to the ONE libamdhip64 of the process (PyTorch
    # bundles its own copy; two HIP/HSA runtimes in one process cannot both drive the GPU).  RCCL is not linked either:
    # umx_shard.hip resolves the handful of nccl* entry points it needs with dlopen/dlsym when a sharded context is
    # initialised, from the librccl the process already holds (torch's) or from ROCm's.
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
