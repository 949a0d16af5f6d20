"""Where does a pack_kernel build write wrong payload words?  (the 7-waves-per-SIMD build.)

Encodes many-block inputs whose longest codes have 17..24 bits (pack_kernel's one-code-per-push form) over
and over with the library named by HUF_LIB_PATH, compares every stream with the oracle's ON THE DEVICE,
and maps each wrong 32-bit word of a bad launch back to the workgroup's tile, wave, lane and the ordinal
of the word among that lane's finished words - plus whether the wrong value is what the LDS stage held
at the same stage index one tile earlier.  Test tooling: uses oracle/ as the checker.

usage: diag_pack.py [seconds | n<launches>] [seed] [max_reports] [tops e.g. 17,24]
       (n2400 = stop after at least 2400 launches spread over 8 inputs instead of after a time)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle

want_launches = int(sys.argv[1][1:]) if len(sys.argv) > 1 and sys.argv[1].startswith("n") else 0
seconds = 1e9 if want_launches else (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_reports = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tops = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [17, 24]
TILE, SPT = 8192, 32


def tree_lens(tree):
    """code length of every leaf of a serialized tree (preorder, -1 = absent child)"""
    lens = np.zeros(256, dtype=np.int64)
    pos = 0
    stack = []          # depths of pending right children
    depth = 0
    # iterative preorder: node := -1 | index node node
    def walk(p, d):
        v = int(tree[p])
        if v == -1:
            return p + 1
        l0 = p + 1
        r0 = walk(l0, d + 1)
        end = walk(r0, d + 1)
        if tree[l0] == -1 and tree[r0] == -1:
            lens[v & 0xff] = d
        return end
    sys.setrecursionlimit(10000)
    walk(0, 0)
    return lens


def analyse(got, want, offs, data, bs, base_mod16, limit=40):
    nb = len(offs) - 1
    rows = []
    summary = {}
    for b in range(nb):
        d0, d1 = int(offs[b]), int(offs[b + 1])
        if np.array_equal(got[d0:d1], want[d0:d1]):
            continue
        a0 = d0 & ~3
        rec_lo = d0 & 3
        blen = int(want[d0:d0 + 8].view(np.uint64)[0])
        tlen = int(want[d0 + 8:d0 + 10].view(np.int16)[0])
        tree = want[d0 + 10:d0 + 10 + 2 * tlen].view(np.int16)
        lens = tree_lens(tree)
        syms = data[b * bs:b * bs + blen]
        cum = np.concatenate([[0], np.cumsum(lens[syms])])
        hdr_end = rec_lo + 10 + 2 * tlen
        bit0 = hdr_end * 8
        unit_start = bit0 + cum[::SPT]                      # first bit of every 32-symbol lane unit (+ end)
        if (len(cum) - 1) % SPT:
            unit_start = np.concatenate([unit_start, [bit0 + cum[-1]]])
        bad = np.nonzero(got[d0:d1] != want[d0:d1])[0] + d0
        words = np.unique((bad - a0) // 4)
        for W in words:
            e = 32 * (int(W) + 1)
            j = int(np.searchsorted(unit_start, e, side="left")) - 1
            if j < 0:
                rows.append(f"  blk {b}: word {W} in the header"); continue
            j = min(j, len(unit_start) - 2)
            t, tid = j // 256, j % 256
            ordinal = int(W) - int(unit_start[j] >> 5)
            nwords = int(unit_start[j + 1] >> 5) - int(unit_start[j] >> 5)
            bitpos_t = bit0 + int(cum[min(TILE * t, len(cum) - 1)])
            w_lo = bitpos_t >> 5
            i_lo = ((base_mod16 + a0 + 4 * w_lo) & 15) >> 2
            idx = i_lo + int(W) - w_lo
            g = int.from_bytes(got[a0 + 4 * W:a0 + 4 * W + 4].tobytes(), "big")
            w = int.from_bytes(want[a0 + 4 * W:a0 + 4 * W + 4].tobytes(), "big")
            stale = ""
            for back in (1, 2):
                if t - back >= 0:
                    bp = bit0 + int(cum[TILE * (t - back)])
                    wl = bp >> 5
                    il = ((base_mod16 + a0 + 4 * wl) & 15) >> 2
                    Wp = wl - il + idx
                    pv = int.from_bytes(want[a0 + 4 * Wp:a0 + 4 * Wp + 4].tobytes(), "big")
                    if pv == g:
                        stale = f" == stage[idx] of tile-{back}"
            n_in = int(unit_start[j]) & 31
            rows.append(f"  blk {b} tile {t} wave {tid >> 6} lane {tid & 63} word#{ordinal}/{nwords} stage_idx {idx} n_in {n_in} "
                        f"got {g:08x} want {w:08x} xor {g ^ w:08x}{stale}")
            key = (ordinal, "stale" if stale else "other")
            summary[key] = summary.get(key, 0) + 1
    return rows[:limit], summary, len(rows)


if __name__ == '__main__':
    import torch
    from libhuffman_amd.codec import GpuCodec
    from libhuffman_amd import _native
    c, o = GpuCodec(0), Oracle()
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = launches = bad = reports = 0
    print("lib", _native.so_path(), flush=True)
    while time.time() < t_end and (not want_launches or launches < want_launches):
        top = int(rng.integers(tops[0], tops[1] + 1))
        bs = int(rng.choice([65536, 65536, 32768, 100000]))
        nb = int(rng.integers(300, 900))
        w = 0.5 ** np.arange(1, top + 1)
        data = rng.choice(top, size=nb * bs, p=w / w.sum()).astype(np.uint8)
        want, offs = o.encode(data, bs, with_offsets=True)
        want_d = torch.from_numpy(want).cuda()
        d = torch.from_numpy(data).cuda()
        out = torch.empty(c.encode_bound(data.size, bs), dtype=torch.uint8, device="cuda")
        t_case = time.time() + 4.0
        case_first = launches
        while (launches - case_first < want_launches // 8 + 1) if want_launches else (time.time() < t_case):
            for rep in range(50):
                stream, _, length = c.encode(d, bs, out=out)
                launches += 1
                if length != want.size or not torch.equal(stream[:length], want_d):
                    bad += 1
                    if reports < max_reports and length == want.size:
                        reports += 1
                        got = stream[:length].cpu().numpy()
                        rows, summary, total = analyse(got, want, offs, data, bs, out.data_ptr() & 15)
                        print(f"MISMATCH launch {launches} top={top} bs={bs} nb={nb}: {total} wrong words; (ordinal, kind) counts {summary}", flush=True)
                        for r in rows:
                            print(r)
                    elif length != want.size:
                        print("LENGTH MISMATCH", length, want.size, flush=True)
        cases += 1
    print("diag_pack", os.path.basename(_native.so_path()), "ok" if bad == 0 else "FAILED", dict(cases=cases, launches=launches, bad=bad), flush=True)
    sys.exit(1 if bad else 0)
