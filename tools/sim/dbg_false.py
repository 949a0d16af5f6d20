import os, sys, struct, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle
o = Oracle(); c = GpuCodec(0)
def handmade(payload, leaves):
    if leaves == 2: tree = [0x0101, 0x41, -1, -1, 0x42, -1, -1]
    else: tree = [0x0103, 0x0101, 0x41, -1, -1, 0x42, -1, -1, 0x0102, 0x43, -1, -1, 0x44, -1, -1]
    per = 8 if leaves == 2 else 4
    return struct.pack("<Qh", per * len(payload), len(tree)) + b"".join(struct.pack("<h", v) for v in tree) + payload
cfg = [int(x) for x in sys.argv[1:6]]
for leaves in (cfg[0],):
  for pay_bytes, at, fake in ((cfg[1], cfg[2], bool(cfg[3])),):
    rng = np.random.default_rng(77 + at)
    pay = bytearray(rng.integers(0, 256, pay_bytes, dtype=np.uint8).tobytes())
    if fake: pay[at:at + 12] = bytes([1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0xff, 0xff])
    tail = o.encode(datagen.zipf255(5 * 65536), 65536)
    head = o.encode(datagen.uniform256(2 * 65536), 65536)
    blk = np.frombuffer(handmade(bytes(pay), leaves), dtype=np.uint8)
    stream = np.concatenate([head, blk, tail])
    per = 8 if leaves == 2 else 4
    cap = 7 * 65536 + per * pay_bytes + 64
    oerr, oout, oused = o.decode(stream, cap, 1025)
    s = torch.from_numpy(stream).cuda()
    for sequential in (bool(cfg[4]),):
        print("leaves", leaves, "pay", pay_bytes, "fake", fake, "sequential", sequential, "oracle", oerr, oout.size, oused, flush=True)
        out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        err, raw, used = c.decode_stream(s, stream.size, stream.size, out, relaxed=True, sequential=sequential)
        print("   ->", err, raw, used, "equal", bool(np.array_equal(out[:raw].cpu().numpy(), oout)), flush=True)
