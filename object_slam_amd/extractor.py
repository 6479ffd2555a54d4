"""ORBextractor — Python mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-110)
over the C ABI (include/oslam_hip.h).  All arithmetic runs in the HIP library."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


class ORBextractor:
    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height,
                 max_batch=1, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_orb_create(C.byref(self.h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                                      width, height, max_batch, device))
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.width, self.height, self.max_batch = width, height, max_batch
        self.cap = self.L.oslam_orb_max_keypoints(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_orb_destroy(self.h)
            self.h = None

    __del__ = close

    # --- getters, reference include/ORBextractor.h:63-83 ---
    def _tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        check(self.L.oslam_orb_get_scale_tables(self.h, ptr(sf), ptr(isf), ptr(s2), ptr(is2), ptr(nf)))
        return sf, isf, s2, is2, nf

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def GetFeaturesPerLevel(self):
        return self._tables()[4]

    def set_blur_rounding(self, sse2):
        check(self.L.oslam_orb_set_blur_rounding(self.h, int(sse2)))

    # --- operator(), reference src/ORBextractor.cc:1043 ---
    def __call__(self, image, mask=None):
        """image: HxW uint8 (host).  Returns (keypoints KP_DTYPE[n], descriptors uint8[n,32])."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        check(self.L.oslam_orb_extract(self.h, ptr(image), w, h, image.strides[0], ptr(kps), ptr(desc), self.cap,
                                       C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    # --- batch mode: images resident in HBM ---
    def extract_batch_device(self, d_ptr, batch, stride, image_stride, stream=None):
        check(self.L.oslam_orb_extract_batch_device(self.h, C.c_void_p(d_ptr), batch, stride, image_stride,
                                                    C.c_void_p(stream or 0)))

    def results_device(self):
        kp, desc, cnt, st = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.L.oslam_orb_results_device(self.h, C.byref(kp), C.byref(desc), C.byref(cnt), C.byref(st)))
        return kp.value, desc.value, cnt.value, st.value

    def fetch(self, b):
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        check(self.L.oslam_orb_fetch(self.h, b, ptr(kps), ptr(desc), self.cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    # --- mvImagePyramid, reference include/ORBextractor.h:85 ---
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        check(self.L.oslam_orb_level_size(self.h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def pyramid_level(self, level, b=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(self.L.oslam_orb_get_pyramid_level(self.h, b, level, ptr(out)))
        return out

    # --- stage outputs for parity tests ---
    def debug_blurred(self, level, b=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(self.L.oslam_orb_debug_get_blurred(self.h, b, level, ptr(out)))
        return out

    def debug_candidates(self, level, b=0):
        cap = 1 << 16
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int(0)
        check(self.L.oslam_orb_debug_get_candidates(self.h, b, level, ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def debug_level_keys(self, level, b=0):
        cap = self.cap
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int(0)
        check(self.L.oslam_orb_debug_get_level_keys(self.h, b, level, ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def algorithmic_bytes(self, n_keypoints):
        return int(self.L.oslam_orb_algorithmic_bytes(self.h, n_keypoints))

    def set_profiling(self, on):
        check(self.L.oslam_orb_set_profiling(self.h, int(on)))

    def get_profile(self):
        """Returns (ms[5] accumulated per kernel group, batches, images)."""
        ms = (C.c_double * 5)()
        nb, ni = C.c_longlong(), C.c_longlong()
        check(self.L.oslam_orb_get_profile(self.h, ms, C.byref(nb), C.byref(ni)))
        return list(ms), nb.value, ni.value
