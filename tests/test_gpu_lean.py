"""The opt-in one-pass decoder of streams that come with the block index alone (kernels/decode_lean.hpp, DESIGN.md 3.5;
HUF_GPU_LEAN_DECODE=1, read when the library decodes for the first time - hence child processes).  It is not the default
(it is slower than decode_fast_kernel), but it is in the library and must give the reference's results: the parity tests
of the index-only path run once more with it switched on, and a sweep of shapes it has special cases for (the stream's
last block, every alignment of the stream, tiny and 1 MiB blocks, deep codes, runs of one byte) against the input."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEP = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
bad = []
def fib(n, k=20):
    f = [1, 1]
    while len(f) < k: f.append(f[-1] + f[-2])
    one = np.concatenate([np.full(v, i, np.uint8) for i, v in enumerate(f)])
    d = np.tile(one, n // one.size + 1)[:n].copy(); np.random.default_rng(5).shuffle(d); return d
def runs(n):
    d = datagen.zipf255(n).copy()
    for at, ln, v in ((1000, 3000, 0), (70000, 20000, 7), (200000, 40, 200)):
        if at + ln < n: d[at:at + ln] = v
    return d
kinds = {"zipf255": datagen.zipf255, "uniform255": datagen.uniform255, "uniform256": datagen.uniform256, "logtext": datagen.logtext,
         "const41": lambda n: datagen.const_bytes(n), "fib": fib, "runs": runs}
handed = 0
for kind, gen in kinds.items():
    for n, bs in ((10, 65536), (3000, 65536), (65536, 65536), (5 * 65536 + 4321, 65536), (300000, 4096), (3 << 20, 1 << 20), (700001, 1 << 18)):
        data_h = gen(n)
        data = torch.from_numpy(data_h).cuda()
        out, offs, length = c.encode(data, bs)
        nb = c.block_count(n, bs)
        for lead in (0, 1, 2, 3):
            big = torch.zeros(length + lead + 64, dtype=torch.uint8, device="cuda")
            big[lead:lead + length] = out[:length]
            back = torch.zeros(n + 7, dtype=torch.uint8, device="cuda")[3:3 + n]          # an output that is not aligned either
            raw = c.decode(big[lead:lead + length], length, offs, nb, back, relaxed=True)
            handed += c.decode_counters()[1]
            if raw != n or not torch.equal(back, data):
                bad.append((kind, n, bs, lead, raw))
print("lean sweep: %%d cases wrong %%s; blocks handed on in all: %%d" %% (len(bad), bad[:5], handed))
sys.exit(1 if bad else 0)
''' % ROOT


def test_the_one_pass_decoder_on_the_shapes_it_has_cases_for(torch_mod_available):
    env = dict(os.environ, HUF_GPU_LEAN_DECODE="1")
    r = subprocess.run([sys.executable, "-c", SWEEP], env=env, capture_output=True, text=True, timeout=900)
    print("\n  " + r.stdout.strip().replace("\n", "\n  "))
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]


def test_the_index_only_parity_tests_with_the_one_pass_decoder_switched_on(torch_mod_available):
    """error codes, delivered bytes after a damaged block, walks that leave the tree, damaged lengths, deep codes, block
    offsets across scan groups: the tests that pin hufgpu_decode() to the oracle, run again with HUF_GPU_LEAN_DECODE=1"""
    env = dict(os.environ, HUF_GPU_LEAN_DECODE="1")
    sel = ("decode_indexed_roundtrip or decode_strict_rejects_k256 or payload_walk_leaves_tree or damaged_block_len or "
           "self_synchronisation or block_offsets_across_groups or block_index_alternating or many_blocks_of_deep_codes or "
           "damaged_stream_matches_the_oracle or index_only_decode")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_subindex.py"), "-m", "gpu", "-x", "-q", "-k", sel],
                       env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-8:])
    print("\n  " + tail.replace("\n", "\n  "))
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail


@pytest.fixture(scope="module")
def torch_mod_available():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch
