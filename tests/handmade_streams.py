"""Blocks no encoder wrote: random trees (any shape src/tree.c:138-227 deserializes, with or without the encoder's root
that has a left child only), random symbols coded with them.  The reference's decoder (src/decoder.c:34-96) walks whatever
tree a stream brings, so its restatement in oracle/ says what every such stream decodes to; the tests compare the HIP
decoders with it.  Test infrastructure (CPU, numpy)."""
from __future__ import annotations

import struct

import numpy as np


def random_tree(rng, leaves: int, skew: float):
    """nested tuples: a leaf is its byte value, a node (left, right); skew = how often a node keeps ONE leaf on a side"""
    syms = [int(x) for x in rng.choice(256, size=leaves, replace=False)]

    def build(items):
        if len(items) == 1:
            return items[0]
        if rng.random() < skew:
            k = 1 if rng.random() < 0.5 else len(items) - 1
        else:
            k = int(rng.integers(1, len(items)))
        return (build(items[:k]), build(items[k:]))
    return build(syms)


def serialize(tree, wrap: bool) -> list:
    """preorder entries as src/tree.c writes them: an index per node (>= 256 for inner nodes), -1 where a child is missing"""
    out, counter = [], [256]

    def rec(t):
        if isinstance(t, int):
            out.extend([t, -1, -1])
        else:
            out.append(counter[0])
            counter[0] += 1
            rec(t[0])
            rec(t[1])
    if wrap:
        out.append(counter[0])
        counter[0] += 1
        rec(tree)
        out.append(-1)
    else:
        rec(tree)
    return out


def codes(tree, wrap: bool) -> dict:
    """byte value -> list of bits"""
    table = {}

    def rec(t, prefix):
        if isinstance(t, int):
            table[t] = prefix
        else:
            rec(t[0], prefix + [0])
            rec(t[1], prefix + [1])
    rec(tree, [0] if wrap else [])
    return table


def block(rng, leaves: int, skew: float, wrap: bool, nsym: int, deep_often: bool = False, pad_ones: bool = False):
    """(bytes of the block, its symbols).  deep_often: every leaf equally likely (long codes all the time) instead of
    likely in proportion to 2^-depth."""
    tree = random_tree(rng, leaves, skew)
    table = codes(tree, wrap)
    keys = sorted(table)
    if deep_often:
        p = np.full(len(keys), 1.0 / len(keys))
    else:
        p = np.array([2.0 ** -min(len(table[k]), 40) for k in keys])
        p /= p.sum()
    syms = np.array(keys, dtype=np.uint8)[rng.choice(len(keys), size=nsym, p=p)]
    bits = np.fromiter((b for s in syms for b in table[int(s)]), dtype=np.uint8)
    pad = (-bits.size) % 8
    if pad:
        bits = np.concatenate([bits, np.full(pad, 1 if pad_ones else 0, dtype=np.uint8)])
    ent = serialize(tree, wrap)
    hdr = struct.pack("<Qh", nsym, len(ent)) + b"".join(struct.pack("<h", v) for v in ent)
    return hdr + np.packbits(bits).tobytes(), syms, max(len(v) for v in table.values())
