"""roctx ranges around the phases of an estimation (statistics / all-reduce / scan / solve), so that a
rocprofv3 trace of a multi-rank job reads as phases and not only as kernel names (the
reference logs wall-clock deltas, DE/fdp.py:94-96, :115-116).

    rocprofv3 --kernel-trace --marker-trace -- python3 bench.py ...      (with DME_ROCTX=1)

The ranges bracket the HOST side of a phase (its launches are enqueued inside the range; the kernels
follow on the stream).  Off unless DME_ROCTX is set to a non-empty value other than 0: a push / pop pair
through ctypes costs about a microsecond of host time per phase.  No range is ever left open: the
context manager pops in a `finally`.
"""
import contextlib
import ctypes
import os

_state = {"lib": None, "tried": False}


def _roctx():
    if not _state["tried"]:
        _state["tried"] = True
        if os.environ.get("DME_ROCTX", "") not in ("", "0"):
            for name in ("librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1",
                         "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.argtypes = []
                    lib.roctxRangePop.restype = ctypes.c_int
                    _state["lib"] = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _state["lib"]


_NULL = contextlib.nullcontext()


class _Range:
    __slots__ = ("lib", "name")

    def __init__(self, lib, name):
        self.lib, self.name = lib, name

    def __enter__(self):
        self.lib.roctxRangePushA(self.name)

    def __exit__(self, *exc):
        self.lib.roctxRangePop()
        return False


def phase(name):
    """Context manager: a roctx range named `dme:<name>` (a no-op unless DME_ROCTX is set)."""
    lib = _roctx()
    if lib is None:
        return _NULL
    return _Range(lib, ("dme:" + name).encode())


def enabled():
    return _roctx() is not None
