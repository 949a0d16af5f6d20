"""CPU check of the rule the decode kernel derives child links from (csrc/kernels/decode.hpp, step 1).

The reference rebuilds the tree recursively from its preorder dump (src/tree.c:138-227: an entry
!= -1 is a node followed by its left and its right subtree, -1 is an absent child, an entry past
the buffer is an absent child too).  The kernel has no recursion: with S(i) = child slots still
open before entry i (S(0) = 1, +1 behind a node, -1 behind a marker) the dump ends where S reaches
0 (`eff`), the left child of node j is entry j+1 and its right child is the first r > j with
S(r) <= S(j).  This test compares that rule with the recursion on valid, truncated and random dumps.
"""
import ctypes as C

import numpy as np

from oracle.oracle import Oracle


def links_by_recursion(buf):
    n = len(buf)
    left, right = {}, {}
    pos = 0
    # iterative preorder parse: stack of (node, side) slots waiting for a subtree
    root = None
    stack = [("root", None)]
    while stack and pos <= n:
        slot = stack.pop()
        if pos >= n:                      # missing entry: NULL child
            continue
        v = buf[pos]
        here = pos
        pos += 1
        if v == -1:
            continue
        if slot[0] == "root":
            root = here
        elif slot[0] == "L":
            left[slot[1]] = here
        else:
            right[slot[1]] = here
        stack.append(("R", here))         # right subtree is parsed after the left one
        stack.append(("L", here))
    return root, left, right


class _DTree(C.Structure):             # hufo_dtree_t (oracle/huf_oracle.h)
    _fields_ = [("left", C.c_int16 * 1026), ("right", C.c_int16 * 1026), ("value", C.c_int16 * 1026), ("n", C.c_int)]


def links_by_oracle(lib, buf):
    """the pinned C restatement of src/tree.c:138-227; its nodes are numbered in creation (= dump) order"""
    arr = (C.c_int16 * max(len(buf), 1))(*buf)
    t = _DTree()
    lib.hufo_tree_deserialize.restype = C.c_size_t
    used = lib.hufo_tree_deserialize(arr, C.c_size_t(len(buf)), C.byref(t))
    at = [i for i in range(used) if buf[i] != -1]          # node number -> entry index
    assert len(at) == t.n
    left = {at[k]: at[t.left[k]] for k in range(t.n) if t.left[k] >= 0}
    right = {at[k]: at[t.right[k]] for k in range(t.n) if t.right[k] >= 0}
    return (at[0] if t.n else None), left, right


def links_by_open_slots(buf):
    n = len(buf)
    S, run, eff = [], 1, n
    for i, v in enumerate(buf):
        if run <= 0:
            eff = i
            break
        S.append(run)
        run += 1 if v != -1 else -1
    left, right = {}, {}
    for j in range(eff):
        if buf[j] == -1:
            continue
        if j + 1 < eff and buf[j + 1] != -1:
            left[j] = j + 1
        r = next((r for r in range(j + 1, eff) if S[r] <= S[j]), eff)
        if r < eff and buf[r] != -1:
            right[j] = r
    root = 0 if eff > 0 and buf[0] != -1 else None
    return root, left, right


def random_dump(rng, max_nodes):
    """preorder dump of a random binary tree shape (children present with probability p)"""
    p = rng.uniform(0.3, 0.95)
    out, stack, nodes = [], [True], 0
    while stack:
        stack.pop()
        if nodes < max_nodes and rng.random() < p:
            out.append(int(rng.integers(0, 512)))
            nodes += 1
            stack += [True, True]
        else:
            out.append(-1)
    return out


def test_child_links_rule_equals_the_recursive_parse():
    rng = np.random.default_rng(11)
    cases = [[], [-1], [7], [7, -1], [256, 65, -1, -1, -1], [1, 2, 3], [1, -1, 2, -1, -1, 9, 9]]
    for _ in range(1500):
        d = random_dump(rng, int(rng.integers(1, 300)))
        cases.append(d)
        cases.append(d[: int(rng.integers(0, len(d) + 1))])                       # cut short
        cases.append(d + [int(x) for x in rng.integers(-1, 5, size=int(rng.integers(0, 6)))])   # trailing entries
        g = [int(x) for x in rng.integers(-1, 3, size=int(rng.integers(0, 40)))]  # anything goes
        cases.append(g)
    lib = Oracle().lib
    for buf in cases:
        want = links_by_recursion(buf)
        assert links_by_open_slots(buf) == want, buf
        assert links_by_oracle(lib, buf) == want, buf
