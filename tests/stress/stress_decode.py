"""Repeats indexed and raw-stream decodes of alternating inputs and checks every output (races in the
decode kernels would show up as rare mismatches).  usage: stress_decode.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
c = GpuCodec(0)
rng = np.random.default_rng(3)
sets = []
for kind, n, bs in (("zipf255", 6 << 20, 65536), ("uniform256", 4 << 20, 65536), ("logtext", 8 << 20, 1 << 20),
                    ("zipf255", 1_000_003, 4097), ("const41", 8 << 20, 65536)):
    d = torch.from_numpy(datagen.GENERATORS[kind](n)).cuda()
    out, offs, ln = c.encode(d, bs)
    sets.append((d, out.clone(), offs.clone(), ln, c.block_count(n, bs)))
# a deep-code input: long codes take the bit-walk path
f = [1, 1]
while sum(f) < 3_000_000: f.append(f[-1] + f[-2])
deep = np.concatenate([np.full(v, i, np.uint8) for i, v in enumerate(f[:40])]); rng.shuffle(deep)
d = torch.from_numpy(deep).cuda(); out, offs, ln = c.encode(d, 1 << 20)
sets.append((d, out.clone(), offs.clone(), ln, c.block_count(deep.size, 1 << 20)))
t_end = time.time() + budget
launches = bad = 0
backs = [torch.empty(s[0].numel() + 64, dtype=torch.uint8, device="cuda") for s in sets]
while time.time() < t_end:
    for i, (d, out, offs, ln, nb) in enumerate(sets):
        n = d.numel()
        raw = c.decode(out, ln, offs, nb, backs[i], relaxed=True)
        ok = raw == n and torch.equal(backs[i][:n], d)
        backs[i].zero_()
        err, raw2, used = c.decode_stream(out, ln, ln, backs[i], relaxed=True)
        ok2 = (err, raw2, used) == (0, n, ln) and torch.equal(backs[i][:n], d)
        launches += 2
        if not (ok and ok2):
            bad += 1
            print("MISMATCH", dict(set=i, indexed=ok, raw=ok2, err=err, raw_len=(raw, raw2, n)), flush=True)
print("stress_decode", "FAILED" if bad else "ok", dict(launches=launches, bad=bad))
