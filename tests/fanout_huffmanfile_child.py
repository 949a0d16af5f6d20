"""Child process of test_gpu_bigraw.py::test_huffmanfile_over_several_sessions: BASELINE configs[4]'s stated route -
log text in 1 MiB blocks through the huffmanfile layer (huffmanfile.py:294-342, 385-417 of the reference) - with
several device sessions configured (HUF_GPU_DEVICES of this process), so that ONE compress() / decompress() call is
dealt out over them (encode_fanout / decode_fanout of csrc/huf_host.cpp).  The stream must be the oracle's byte for
byte, the round trip the input, and both fan-out counters must have moved."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, sys.argv[1])
from libhuffman_amd import _native as N, datagen, huffmanfile  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

mib = int(sys.argv[2]) if len(sys.argv) > 2 else 160
bs = 1 << 20
n = (mib << 20) + 4321
data = datagen.logtext(n).tobytes()
L = N.load()
want = Oracle().encode(np.frombuffer(data, np.uint8), bs)

fe0, fd0 = C.c_int(0), C.c_int(0)
L.huf_gpu_fanouts(C.byref(fe0), C.byref(fd0))
t0 = time.perf_counter()
enc = huffmanfile.compress(data, blocksize=bs)
t1 = time.perf_counter()
back = huffmanfile.decompress(enc)
t2 = time.perf_counter()
fe, fd = C.c_int(0), C.c_int(0)
L.huf_gpu_fanouts(C.byref(fe), C.byref(fd))
configured = C.c_int(0)
live = L.huf_gpu_sessions(C.byref(configured))
ok_stream = len(enc) == want.size and np.array_equal(np.frombuffer(enc, np.uint8), want)
ok_back = back == data
print(f"sessions live={live} configured={configured.value} MiB={mib} compress_s={t1 - t0:.3f} decompress_s={t2 - t1:.3f} "
      f"fanout_encodes={fe.value - fe0.value} fanout_decodes={fd.value - fd0.value} stream_is_oracles={ok_stream} roundtrip={ok_back}")
# the incremental objects take the same route
comp = huffmanfile.HuffmanCompressor(blocksize=bs)
half = (n // 2) // bs * bs
enc2 = comp.compress(data[:half]) + comp.compress(data[half:]) + comp.flush()
ok_inc = enc2 == enc
print(f"incremental_equals_one_shot={ok_inc}")
sys.exit(0 if (ok_stream and ok_back and ok_inc) else 1)
