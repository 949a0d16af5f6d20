"""Deterministic synthetic inputs for the benchmark configurations.

Integer-only, counter-based ``splitmix64`` so that the same bytes can be produced by numpy
(here), by C, and by the HIP fill kernels in ``csrc/hufgpu_datagen.hip`` (the device
generators are checked against this module in tests/test_gpu_parity.py).

    z_i   = seed + i * 0x9E3779B97F4A7C15                (i >= 1, mod 2**64)
    z     = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
    z     = (z ^ (z >> 27)) * 0x94D049BB133111EB
    out_i = z ^ (z >> 31)
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, count: int, first: int = 1) -> np.ndarray:
    """Outputs ``first .. first+count-1`` of the counter-based splitmix64 stream."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + count, dtype=np.uint64)
        z = np.uint64(seed) + i * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def const_bytes(n: int, value: int = 0x41) -> np.ndarray:
    """BASELINE config 2: ``n`` copies of one byte (degenerate one-symbol tree)."""
    return np.full(n, value, dtype=np.uint8)


def uniform256(n: int, seed: int = 1) -> np.ndarray:
    """BASELINE config 4: raw little-endian bytes of successive splitmix64 outputs."""
    words = splitmix64(seed, (n + 7) // 8)
    return words.astype("<u8").view(np.uint8)[:n].copy()


def uniform255(n: int, seed: int = 2) -> np.ndarray:
    """Config 4b: ``splitmix64_j mod 255`` - byte 255 never occurs, so k <= 255."""
    return (splitmix64(seed, n) % np.uint64(255)).astype(np.uint8)


def zipf_cumulative(symbols: int = 255) -> np.ndarray:
    r = np.arange(1, symbols + 1, dtype=np.uint64)
    w = np.uint64(1 << 32) // r
    return np.cumsum(w, dtype=np.uint64)


def zipf255(n: int, seed: int = 3) -> np.ndarray:
    """BASELINE config 3: Zipf(s=1) over byte values 0..254 (255 never occurs).

    ``byte_j`` = number of cumulative weights ``C_r <= splitmix64_j mod W``.
    """
    cum = zipf_cumulative(255)
    u = splitmix64(seed, n) % cum[-1]
    return np.searchsorted(cum, u, side="right").astype(np.uint8)


_LOG_LEVELS = (b"INFO", b"WARN", b"DEBUG", b"ERROR", b"TRACE")
_LOG_COMPONENTS = (b"scheduler", b"net.rpc", b"storage.io", b"auth", b"gc", b"kernel", b"http.api", b"cache")
_LOG_MESSAGES = (
    b"request completed in %d us status=%d",
    b"connection from 10.%d.%d.7 accepted",
    b"flushed %d pages to segment 0x%x",
    b"retrying operation id=%x attempt=%d",
    b"cache miss for key %x (%d bytes)",
    b"heartbeat ok seq=%d lag=%d ms",
)


def logtext(n: int, seed: int = 5) -> np.ndarray:
    """BASELINE config 5: synthetic log lines (printable ASCII + newline, k < 100).

    Lines are ``2026-10-03T12:MM:SS.mmmZ LEVEL component: message`` with numeric fields drawn
    from splitmix64.  A fixed set of 4096 lines is generated and tiled, with the line order
    permuted per 4096-line page by the PRNG, which keeps generation fast at 16 GiB.
    """
    rnd = splitmix64(seed, 4096 * 4).reshape(4096, 4)
    lines = []
    for a, b, c, d in rnd.tolist():
        msg = _LOG_MESSAGES[a % len(_LOG_MESSAGES)] % ((b >> 8) % 100000, (c >> 8) % 4096)
        lines.append(
            b"2026-10-03T12:%02d:%02d.%03dZ %s %s: %s\n"
            % (a >> 8 & 0x3F if (a >> 8 & 0x3F) < 60 else 59, b & 0x3F if (b & 0x3F) < 60 else 59,
               c % 1000, _LOG_LEVELS[d % len(_LOG_LEVELS)],
               _LOG_COMPONENTS[(d >> 8) % len(_LOG_COMPONENTS)], msg)
        )
    page = np.frombuffer(b"".join(lines), dtype=np.uint8)
    reps = n // page.size + 1
    return np.tile(page, reps)[:n].copy()


GENERATORS = {
    "const41": lambda n: const_bytes(n, 0x41),
    "uniform256": lambda n: uniform256(n, 1),
    "uniform255": lambda n: uniform255(n, 2),
    "zipf255": lambda n: zipf255(n, 3),
    "logtext": lambda n: logtext(n, 5),
}
