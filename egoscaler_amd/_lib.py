"""ctypes binding of libegomi.so (the C-ABI declared in include/egomi.h).

There is NO fallback: if the library is missing or a symbol is absent this module raises, and every
op in egoscaler_amd.ops raises with it.  Nothing here imports oracle/.
"""
import ctypes
import os

# torch must be imported BEFORE libegomi.so is dlopen'ed: the PyTorch-ROCm wheel carries its own
# libamdhip64; loading ours first would bring a second HIP runtime into the process (the system
# one), and launches through it fail with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libegomi.so")
_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_d = ctypes.c_double
c_f = ctypes.c_float
c_sz = ctypes.c_size_t
c_i64 = ctypes.c_int64


class EgomiError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EgomiError(f"{LIB_PATH} not found: build it with `python -m egoscaler_amd.build` "
                             "(the product path has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.egomi_strerror.restype = ctypes.c_char_p
        _lib.egomi_strerror.argtypes = [c_i]
        _lib.egomi_last_launch_error.restype = ctypes.c_char_p
        _lib.egomi_unproject_workspace_bytes.restype = c_sz
        _lib.egomi_unproject_workspace_bytes.argtypes = [c_i] * 4
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().egomi_strerror(rc).decode()
        if rc == -3:
            msg += ": " + lib().egomi_last_launch_error().decode()
        raise EgomiError(f"{what}: egomi error {rc} ({msg})")


_SYNC_DEBUG = os.environ.get("EGOMI_SYNC_DEBUG")     # debug aid: a file; every library call is followed by a device synchronisation, and the first call
                                                      # after which the device reports an error is written there with its Python stack (a fault pinned to its launch)


def call(name: str, *args):
    fn = getattr(lib(), name)
    if not _SYNC_DEBUG:
        check(fn(*args), name)
        return
    import traceback
    import torch
    try:
        check(fn(*args), name)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
    except BaseException as e:                                        # noqa: BLE001
        with open(_SYNC_DEBUG, "a") as f:
            f.write(f"=== {name} raised {type(e).__name__}: {str(e)[:200]}\n")
            f.write("".join(traceback.format_stack(limit=14)))
            f.write(f"args: {[getattr(a, 'value', a) for a in args]}\n")
        raise
