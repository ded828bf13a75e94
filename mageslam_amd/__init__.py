"""mageslam_amd -- MI355X-native back-end for the data-parallel hot path of microsoft/mageslam.

The product is the shared library ``libmageslam_hip.so`` (hand-written HIP for gfx950 behind the C ABI
of ``include/*.h``) plus the C++ shim ``include/BundlerLib.h``.  This package only holds the build
driver, thin ctypes bindings that mirror the reference's operator surface, and the synthetic-scene
generator used by tests and bench.py.  There is no CPU fallback anywhere in this package.
"""
__all__ = ["scene", "bundler", "build"]
