"""Drop-in import name of the reference's Python package: ``import huffmanfile`` gives the
GPU-backed implementation of libhuffman_amd.huffmanfile (same public names)."""
from libhuffman_amd.huffmanfile import (DEFAULT_BLOCK_SIZE, DEFAULT_MEM_LIMIT, HuffmanCompressor,  # noqa: F401
                                        HuffmanDecompressor, HuffmanError, HuffmanFile, compress,
                                        decompress, open)

__all__ = ["HuffmanError", "HuffmanFile", "HuffmanCompressor", "HuffmanDecompressor",
           "compress", "decompress", "open"]
