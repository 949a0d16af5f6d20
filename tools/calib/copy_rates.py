"""Host <-> device copy rates on this box: pageable and pinned, one direction and both at once.
What huf_encode()/huf_decode() on host memory streams can reach at best."""
import threading
import time

import numpy as np
import torch

n = 256 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
page = torch.from_numpy(np.ones(n, np.uint8))
page2 = torch.from_numpy(np.ones(n, np.uint8))
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin2 = torch.empty(n, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def rate(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t0) / 2**30


print(f"H2D pageable {rate(lambda: dev.copy_(page)):.1f} GiB/s, pinned {rate(lambda: dev.copy_(pin, non_blocking=True)):.1f}")
print(f"D2H pageable {rate(lambda: page.copy_(dev)):.1f} GiB/s, pinned {rate(lambda: pin.copy_(dev, non_blocking=True)):.1f}")


def both_pinned():
    with torch.cuda.stream(s1):
        dev.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2):
        pin2.copy_(dev2, non_blocking=True)


print(f"pinned both directions at once: {2 * rate(both_pinned):.1f} GiB/s in total")


def both_pageable():
    def up():
        with torch.cuda.stream(s1):
            dev.copy_(page)
            s1.synchronize()
    def down():
        with torch.cuda.stream(s2):
            page2.copy_(dev2)
            s2.synchronize()
    a, b = threading.Thread(target=up), threading.Thread(target=down)
    a.start(); b.start(); a.join(); b.join()


print(f"pageable both directions (two threads): {2 * rate(both_pageable):.1f} GiB/s in total")
t0 = time.perf_counter()
page2.copy_(page)
print(f"host memcpy one thread: {n / (time.perf_counter() - t0) / 2**30:.1f} GiB/s")
