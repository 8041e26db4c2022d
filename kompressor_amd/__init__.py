"""kompressor_amd -- MI355X (gfx950) backend for Kompressor's zstd hot path.

Only what the path needs: the HIP kernels + C ABI (csrc/, include/kompressor_hip.h),
the host-side mirror of the reference's SliceTransform / ZstdCompressor /
ZstdDecompressor interface, the batched device API, and the seeded corpus."""
from .slice_transform import ByteArraySlice, SliceTransform  # noqa: F401


def __getattr__(name):
    # codecs load the HIP library on first use and fail loudly if it is missing
    if name in ("ZstdCompressor", "ZstdDecompressor"):
        from . import zstd
        return getattr(zstd, name)
    if name in ("ZlibCompressor", "ZlibDecompressor", "ZlibFormat"):
        from . import zlib
        return getattr(zlib, name)
    if name in ("ZstdBatch", "compress_bound"):
        from . import batch
        return getattr(batch, name)
    raise AttributeError(name)
