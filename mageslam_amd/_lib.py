"""Loader of libmageslam_hip.so (the C ABI declared in include/*.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded this raises, and every
compute entry point returns MAGE_ERR_NO_DEVICE when no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmageslam_hip.so")

MAGE_OK, MAGE_ERR_INVALID_ARGUMENT, MAGE_ERR_OUT_OF_MEMORY, MAGE_ERR_DEVICE, MAGE_ERR_UNSUPPORTED, MAGE_ERR_NO_DEVICE = range(6)
_STATUS_NAMES = {0: "MAGE_OK", 1: "MAGE_ERR_INVALID_ARGUMENT", 2: "MAGE_ERR_OUT_OF_MEMORY", 3: "MAGE_ERR_DEVICE",
                 4: "MAGE_ERR_UNSUPPORTED", 5: "MAGE_ERR_NO_DEVICE"}


class MageError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{_STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m mageslam_amd.build` (hipcc, gfx950). "
                "mageslam_amd has no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP / HSA runtime.  In a process that uses both (tests, bench.py, the Python
        # window driver) torch's runtime has to come up FIRST: when this library's (system) runtime initialises first, torch
        # later reports "No HIP GPUs are available".  Harness detail only -- the shared library itself never touches torch.
        import sys
        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available():
                    torch.cuda.init()
            except Exception:            # noqa: a CPU-only torch is fine
                pass
        _lib = C.CDLL(LIB_PATH)
        _lib.mage_last_error.restype = C.c_char_p
    return _lib


def check(status: int) -> None:
    if status != MAGE_OK:
        raise MageError(status, lib().mage_last_error().decode("utf-8", "replace"))
