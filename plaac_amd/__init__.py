"""plaac_amd — MI355X-native PLAAC scoring engine.

The product is the C-ABI shared library `libplaac_native.so` (hand-written HIP kernels for gfx950,
see include/plaac_native.h). This package is the thin Python host mirror of that ABI:
    native   ctypes binding (Context, make_params, encode, pack, ROW_DTYPE ...)
    synth    seeded synthetic proteomes for the BASELINE configs
    dist     one-process-per-GPU sharding + final gather of summary rows (torch.distributed)
Nothing here computes scores on the CPU; without the built library importing `native.load()` raises.
"""
from . import native  # noqa: F401

__all__ = ["native"]
