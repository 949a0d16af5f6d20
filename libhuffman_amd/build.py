"""Builds libhuffman_amd/libhuffman.so: HIP kernels + device C ABI + host drop-in API.

    python -m libhuffman_amd.build          (or __graft_entry__.build())

hipcc cross-compiles for gfx950 without a GPU.  The .so is built in-tree so that it travels
with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import signal
import subprocess
import sys
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SO_PATH = os.environ.get("HUF_LIB_PATH") or os.path.join(PKG, "libhuffman.so")   # override: tooling experiments only
SOURCES = ["hufgpu_api.hip", "huf_host.cpp", "hufgpu_sharded.hip"]
KERNEL_PARTS = ["util", "histogram", "tree", "offsets", "hist_tree", "hist_lanes", "pack", "hist_chunk", "pack_chunk", "decode", "decode_sub", "decode_fast", "decode_regs", "spec_index", "discover", "fill"]
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


def device_asm(extra_flags=None) -> str:
    """gfx950 assembly of the kernel translation unit, compiled with the flags of the library build"""
    flags = os.environ.get("HUF_EXTRA_FLAGS", "").split() if extra_flags is None else list(extra_flags)
    flags = [f for f in flags if not f.startswith("-fsanitize")]          # host-only instrumentation
    with tempfile.TemporaryDirectory(prefix="hufasm") as tmp:
        out = os.path.join(tmp, "kernels.s")
        subprocess.check_call([_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                               "-Wno-unused-command-line-argument"] + flags +
                              ["-x", "hip", os.path.join(CSRC, "hufgpu_api.hip"), "-o", out])
        with open(out) as f:
            return f.read()


def check_isa(extra_flags=None) -> dict:
    """Raises RuntimeError when a kernel carries the gfx950 last-VGPR shift hazard (isa_check.py);
    returns the per-kernel register / LDS table otherwise."""
    from . import isa_check
    asm = device_asm(extra_flags)
    hazards = isa_check.last_vgpr_shift_hazards(asm)
    if hazards:
        raise RuntimeError(isa_check.format_hazards(hazards))
    return isa_check.kernel_resources(asm)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO_PATH
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function"] + os.environ.get("HUF_EXTRA_FLAGS", "").split() + [   # tooling experiments only
           "-x", "hip", os.path.join(CSRC, "hufgpu_api.hip"),
           "-x", "hip", os.path.join(CSRC, "huf_host.cpp"),
           "-x", "hip", os.path.join(CSRC, "hufgpu_sharded.hip"),
           "-o", SO_PATH + ".tmp", "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    main = subprocess.Popen(cmd, start_new_session=True)        # own process group: hipcc's children die with it
    try:
        # the ISA of the same kernels, checked while the library links (HUF_SKIP_ISA_CHECK=1: only for
        # experiments that WANT a hazardous build, e.g. tools/diag_pack.py's reproducer)
        if os.environ.get("HUF_SKIP_ISA_CHECK") != "1":
            check_isa()
    except BaseException:
        try:
            os.killpg(main.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        main.wait()
        if os.path.exists(SO_PATH + ".tmp"):
            os.remove(SO_PATH + ".tmp")
        raise
    if main.wait() != 0:
        raise subprocess.CalledProcessError(main.returncode, cmd)
    os.replace(SO_PATH + ".tmp", SO_PATH)
    return SO_PATH


if __name__ == "__main__":
    if "--resources" in sys.argv:
        for name, r in check_isa().items():
            print(f"{r['vgprs']:4d} vgprs ({r['allocated']:3d} allocated) {r['sgprs']:4d} sgprs {r['lds']:6d} B LDS {r['scratch']:5d} B scratch  {name}")
    else:
        print(build(force="--force" in sys.argv, verbose=True))
