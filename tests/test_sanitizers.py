"""The host side of the library (huf_host.cpp: memstreams, fd streams, bufio, the tree / histogram / symbol
API, argument checks, the session pool's device list) under AddressSanitizer + UndefinedBehaviorSanitizer.

The reference's only memory check is valgrind over every test (test/CMakeLists.txt:8-26); valgrind is not in
this image, clang's sanitizers are.  The library is rebuilt with -fsanitize=address,undefined (host code only:
hipcc ignores the flag for gfx950, and GPU ASan is not available on this pool) and tests/test_abi.py runs
against it in a child process with the ASan runtime preloaded.  Any report fails the child
(halt_on_error); leak checking is off because CPython itself never frees everything.
"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asan_runtime():
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        return None
    path = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if os.path.isabs(path) and os.path.exists(path):
        return path
    hits = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


def test_host_api_under_asan_and_ubsan():
    rt = asan_runtime()
    if rt is None:
        pytest.skip("clang's ASan runtime is not in this image")
    lib = os.path.join(ROOT, "libhuffman_amd", "_variants", "asan_ubsan.so")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    env = dict(os.environ, HUF_LIB_PATH=lib, HUF_EXTRA_FLAGS="-fsanitize=address,undefined -fno-omit-frame-pointer -g")
    subprocess.check_call([sys.executable, "-m", "libhuffman_amd.build"], cwd=ROOT, env=env)
    env = dict(os.environ, HUF_LIB_PATH=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("HUF_EXTRA_FLAGS", None)
    # (the cffi test starts /opt/conda's python, whose libstdc++ does not go with a preloaded runtime)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi.py"), "-q", "-x", "-m", "not gpu",
                        "-k", "not cffi", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    out = r.stdout + r.stderr
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and " passed" in out, out[-4000:]
