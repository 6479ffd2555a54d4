"""ctypes loader of the in-tree gfx950 library (object_slam_amd/liboslam_hip.so).

There is no CPU fallback: if the library is missing or no HIP device is visible the product
entry points raise.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSLAM_LIB_PATH") or os.path.join(HERE, "liboslam_hip.so")   # override: kernel experiments only

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

OSLAM_E_INVALID, OSLAM_E_HIP, OSLAM_E_CAPACITY, OSLAM_E_NUMERIC = -1, -2, -3, -4


class OslamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oslam error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OslamError(OSLAM_E_HIP, "%s not built: run python -m object_slam_amd.build "
                             "(the HIP path has no CPU fallback)" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1; two HIP runtimes in one
        # process cannot both open the GPU, so let torch's load first and share it (same soname).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.oslam_last_error.restype = C.c_char_p
        L.oslam_orb_algorithmic_bytes.restype = C.c_int64
        L.oslam_orb_algorithmic_bytes.argtypes = [C.c_void_p, C.c_int]
        L.oslam_orb_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int]
        L.oslam_orb_destroy.argtypes = [C.c_void_p]
        L.oslam_orb_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise OslamError(rc, lib().oslam_last_error().decode())


def ptr(a):
    # ndarray.ctypes builds a helper object per call (~15 us); the array interface gives the address directly
    return C.c_void_p(a.__array_interface__["data"][0])
