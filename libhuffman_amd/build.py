"""Builds libhuffman_amd/libhuffman.so: HIP kernels + device C ABI + host drop-in API.

    python -m libhuffman_amd.build          (or __graft_entry__.build())

hipcc cross-compiles for gfx950 without a GPU.  The .so is built in-tree so that it travels
with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SO_PATH = os.environ.get("HUF_LIB_PATH") or os.path.join(PKG, "libhuffman.so")   # override: tooling experiments only
SOURCES = ["hufgpu_api.hip", "huf_host.cpp"]
KERNEL_PARTS = ["util", "histogram", "tree", "offsets", "hist_tree", "pack", "hist_chunk", "pack_chunk", "decode", "decode_sub", "discover", "fill"]
DEPENDS = SOURCES + ["hufgpu_kernels.hip", "hufgpu_common.h",
                     os.path.join("..", "..", "include", "huffman.h"),
                     os.path.join("..", "..", "include", "huffman_gpu.h")] + \
          [os.path.join("kernels", part + ".hpp") for part in KERNEL_PARTS]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the codec cannot be built without the ROCm toolchain")


def needs_build() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    built = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > built for d in DEPENDS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO_PATH
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function"] + os.environ.get("HUF_EXTRA_FLAGS", "").split() + [   # tooling experiments only
           "-x", "hip", os.path.join(CSRC, "hufgpu_api.hip"),
           "-x", "hip", os.path.join(CSRC, "huf_host.cpp"),
           "-o", SO_PATH + ".tmp", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(SO_PATH + ".tmp", SO_PATH)
    return SO_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
