"""Drop-in import name of the reference's Python package: ``import huffmanfile`` gives the
GPU-backed implementation that lives in libhuffman_amd.huffmanfile (same public names)."""
from libhuffman_amd import huffmanfile as _impl
from libhuffman_amd.huffmanfile import *  # noqa: F401,F403
from libhuffman_amd.huffmanfile import DEFAULT_BLOCK_SIZE, DEFAULT_MEM_LIMIT  # noqa: F401

__all__ = list(_impl.__all__)
