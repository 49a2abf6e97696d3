"""ctypes loader for the C-ABI kernel library (include/pase_amd.h).

The product path loads pase_amd/libpase_hip.so (hipcc, gfx950) and refuses to run without it:
there is NO CPU fallback.  `use_library()` exists only so the kernel tests can point the very same
Python wrappers at the SIMT-emulator build of the same sources (tests/hipemu) on a GPU-less box.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(_HERE, "libpase_hip.so")

_lib = None
_device_type = "cuda"
LIBRARY_OVERRIDE = None      # path of a PASE_LIB A/B build when one was loaded instead of pase_amd/libpase_hip.so


class PaseLibraryError(RuntimeError):
    pass


def use_library(path, device_type):
    """TEST HOOK: bind the wrappers to another build of the same C ABI (the CPU emulator)."""
    global _lib, _device_type
    _lib = ctypes.CDLL(path) if path is not None else None
    _device_type = device_type
    if _lib is not None:
        _declare(_lib)


def device_type():
    return _device_type


def lib():
    global _lib
    if _lib is None:
        ab = os.environ.get("PASE_LIB")
        if ab:
            # A/B MEASUREMENT builds of the same sources with another -D flag (tools/ab_build.sh); never set in production.
            # No source-digest check is possible (the flags differ by construction): the override is loud instead -- printed
            # once, and bench.py records LIBRARY_OVERRIDE in its JSON line
            global LIBRARY_OVERRIDE
            LIBRARY_OVERRIDE = os.path.abspath(ab)
            import sys
            sys.stderr.write("pase_amd: PASE_LIB override, loading %s (A/B measurement build, no freshness check)\n" % LIBRARY_OVERRIDE)
            _lib = ctypes.CDLL(ab)
            _declare(_lib)
            return _lib
        if not os.path.exists(HIP_SO):
            raise PaseLibraryError(
                "pase_amd: %s is missing. Build it with `python -m pase_amd.build hip` "
                "(or __graft_entry__.build()); there is no CPU fallback." % HIP_SO)
        _check_fresh()
        _lib = ctypes.CDLL(HIP_SO)
        _declare(_lib)
    return _lib


def _check_fresh():
    """A stale binary with the same ABI but older kernel semantics would load silently: compare the build
    stamp with the digest of the current sources (pase_amd/build.py) and refuse to run on a mismatch."""
    from . import build
    stamp = HIP_SO + ".sha256"
    if not os.path.exists(stamp):
        return                      # a hand-built library: nothing to compare against
    want = build.hip_digest()
    if open(stamp).read().strip() != want:
        raise PaseLibraryError("pase_amd: %s is older than pase_amd/csrc (source digest mismatch); rebuild with "
                               "`python -m pase_amd.build hip`" % HIP_SO)


def _declare(l):
    from . import kernels
    kernels.declare(l)
