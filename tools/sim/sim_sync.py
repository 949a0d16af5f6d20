"""How often does a lane that starts RUNIN bits in front of its share stand on a codeword start when it
reaches the share?  CPU model of the speculation decode_lean relies on (no GPU): one block of each workload,
shares of ~32 symbols, the in-order boundaries as the truth.  Test/tooling only (uses the oracle's encoder)."""
import sys, bisect
import numpy as np
sys.path.insert(0, ".")
from libhuffman_amd import datagen
from oracle.oracle import Oracle

def parse_block(stream, off):
    blen = int.from_bytes(stream[off:off + 8].tobytes(), "little")
    tlen = int.from_bytes(stream[off + 8:off + 10].tobytes(), "little", signed=True)
    tree = np.frombuffer(stream[off + 10:off + 10 + 2 * tlen].tobytes(), dtype="<i2")
    return blen, tlen, tree, off + 10 + 2 * tlen

def codes_from_tree(tree):
    """(code left-aligned in 32 bits, len, sym) of the leaves in preorder"""
    out = []
    pos = 0
    def walk(depth, code):
        nonlocal pos
        v = int(tree[pos]); pos += 1
        if v == -1:
            return False
        p0 = pos
        l = walk(depth + 1, code << 1)
        r = walk(depth + 1, (code << 1) | 1)
        if not l and not r:
            out.append((code << (32 - depth), depth, v & 0xff))
        return True
    sys.setrecursionlimit(10000)
    walk(0, 0)
    return out

def main():
    o = Oracle()
    n, bs = 1 << 16, 1 << 16
    wl = sys.argv[1:] or ["zipf255", "uniform255", "uniform256", "logtext"]
    for kind in wl:
        data = datagen.GENERATORS[kind](n * 3)[n * 2: n * 3]     # third block of the stream
        st = o.encode(data, bs)
        blen, tlen, tree, pay0 = parse_block(st, 0)
        leaves = codes_from_tree(tree)
        codes = [c for c, _, _ in leaves]
        lens = [l for _, l, _ in leaves]
        lenof = {s: l for _, l, s in leaves}
        bits = np.unpackbits(st[pay0:]).astype(np.uint8)
        bits = np.concatenate([bits, np.zeros(4096, np.uint8)])
        truth = np.cumsum([0] + [lenof[int(b)] for b in data])          # bit position of every symbol
        pay_bits = int(truth[-1])
        isb = np.zeros(pay_bits + 8192, bool); isb[truth] = True
        pw = 1 << np.arange(31, -1, -1, dtype=np.uint64)
        def step(p):
            """position behind the codeword (or the failing run) at p"""
            if bits[p]:
                q = p
                while bits[q]: q += 1
                return q
            w = int((bits[p:p + 32].astype(np.uint64) * pw[:len(bits[p:p+32])]).sum())
            k = bisect.bisect_right(codes, w) - 1
            return p + lens[k]
        nshare = (blen + 31) // 32
        sb = -(-pay_bits // nshare)
        print(f"{kind}: {pay_bits / blen:.2f} bits/symbol, K={len(leaves)}, lens {min(lens)}..{max(lens)}, shares of {sb} bits")
        for runin in (0, 32, 64, 96, 128, 192):
            bad = 0; badwave = set(); n_in_share_sync = 0
            for i in range(1, nshare):
                lo = i * sb
                p = max(lo - runin, 0)
                while p < lo: p = step(p)
                if not isb[p]:
                    bad += 1; badwave.add(i // 64)
                    # does it fall into step inside its share?
                    hi = lo + sb
                    while p < hi and not isb[p]: p = step(p)
                    if p < hi or isb[p]: n_in_share_sync += 1
            print(f"   run-in {runin:3d}: {100.0 * bad / nshare:5.1f} % of lanes off at their share's start, "
                  f"{len(badwave)}/{(nshare + 63) // 64} waves hold one; {n_in_share_sync}/{bad} of them in step by the share's end")

main()
