"""Python front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Two interchangeable back-ends with the same methods:

  Oracle("port")       oracle/libgs_oracle.so  -- this repo's C restatement (gs_oracle.c)
  Oracle("reference")  oracle/_ref/libgs_ref.so -- the unmodified reference header compiled
                       by oracle/Makefile (present in the build container and shipped to
                       the GPU box as a prebuilt, git-ignored binary)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from grayskull_amd._abi import (GsImage, GsLbpCascade, KEYPOINT_DTYPE, MATCH_DTYPE, RECT_DTYPE)

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libgs_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libgs_ref.so")
REF_NOSTDLIB_SO = os.path.join(HERE, "_ref", "libgs_ref_nostdlib.so")  # same header, -DGS_NO_STDLIB (ref :68-88)

u8p = C.POINTER(C.c_uint8)


def build():
    """Compile the C restatement (and, when /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def have_reference():
    return os.path.exists(REF_SO)


def have_reference_nostdlib():
    return os.path.exists(REF_NOSTDLIB_SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class GsRect(C.Structure):  # grayskull.h:24
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("w", C.c_uint), ("h", C.c_uint)]


class GsPoint(C.Structure):  # grayskull.h:19
    _fields_ = [("x", C.c_uint), ("y", C.c_uint)]


def _img(a):
    h, w = a.shape
    return GsImage(w, h, a.ctypes.data)


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        if kind == "port":
            if not os.path.exists(PORT_SO):
                build()
            self.lib = C.CDLL(PORT_SO)
        elif kind == "reference":
            self.lib = C.CDLL(REF_SO)
        elif kind == "reference_nostdlib":  # polynomial gs_atan2 / gs_sin (ref :70-88), everything else identical
            self.lib = C.CDLL(REF_NOSTDLIB_SO)
        elif kind == "port_nostdlib":       # the restatement with its trig switched to the same polynomials
            if not os.path.exists(PORT_SO):
                build()
            # a private copy of the library: orc_set_nostdlib is process-global state of one loaded image
            import shutil
            import tempfile
            tmp = tempfile.NamedTemporaryFile(suffix=".so", delete=False)
            tmp.close()
            shutil.copyfile(PORT_SO, tmp.name)
            self.lib = C.CDLL(tmp.name)
            self.lib.orc_set_nostdlib(1)
            self.lib.orc_atan2_poly.restype = C.c_float
            self.lib.orc_sin_poly.restype = C.c_float
        else:
            raise ValueError(kind)
        self.port = kind.startswith("port")
        L = self.lib
        if self.port:
            L.orc_fnv1a.restype = C.c_uint32
            L.orc_fnv1a.argtypes = [C.c_void_p, C.c_uint64]
            L.orc_otsu_threshold.restype = C.c_uint8
            L.orc_otsu_from_hist.restype = C.c_uint8
            L.orc_orientation.restype = C.c_float
            L.orc_lbp_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p,
                                         C.c_uint, C.c_float, C.c_float, C.c_float, C.c_int]
            L.orc_lbp_window.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int,
                                         C.c_int, C.c_float]
            L.orc_match_orb.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p,
                                        C.c_uint, C.c_float]
        else:
            L.gs_otsu_threshold.restype = C.c_uint8
            L.gs_otsu_threshold.argtypes = [GsImage]
            L.gs_compute_orientation.restype = C.c_float
            L.gs_compute_orientation.argtypes = [GsImage, C.c_uint, C.c_uint, C.c_uint]
            L.gs_lbp_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p,
                                        C.c_uint, C.c_float, C.c_float, C.c_float, C.c_int]
            L.gs_lbp_window.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int,
                                        C.c_int, C.c_float]
            L.gs_match_orb.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p,
                                       C.c_uint, C.c_float]
            L.gs_blur.argtypes = [GsImage, GsImage, C.c_uint]
            for f in ("gs_sobel", "gs_erode", "gs_dilate", "gs_downsample"):
                getattr(L, f).argtypes = [GsImage, GsImage]
            L.gs_histogram.argtypes = [GsImage, C.c_void_p]
            L.gs_threshold.argtypes = [GsImage, C.c_uint8]
            L.gs_adaptive_threshold.argtypes = [GsImage, GsImage, C.c_uint, C.c_int]
            L.gs_filter.argtypes = [GsImage, GsImage, GsImage, C.c_uint]
            L.gs_integral.argtypes = [GsImage, C.c_void_p]
            L.gs_fast.argtypes = [GsImage, GsImage, C.c_void_p, C.c_uint, C.c_uint]
            L.gs_brief_descriptor.argtypes = [GsImage, C.c_void_p]
            L.gs_orb_extract.argtypes = [GsImage, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
            L.ref_atan2.restype = C.c_float
            L.ref_atan2.argtypes = [C.c_float, C.c_float]
            L.ref_sin.restype = C.c_float
            L.ref_sin.argtypes = [C.c_float]

    # ------------------------------------------------------------ helpers
    @staticmethod
    def synth(w, h, seed):
        lib = Oracle._port_lib()
        a = np.empty((h, w), np.uint8)
        lib.orc_synth(_p(a), C.c_uint(w), C.c_uint(h), C.c_uint32(seed))
        return a

    @staticmethod
    def fnv1a(a):
        lib = Oracle._port_lib()
        a = np.ascontiguousarray(a)
        return int(lib.orc_fnv1a(_p(a), a.nbytes))

    _port = None

    @staticmethod
    def _port_lib():
        if Oracle._port is None:
            Oracle._port = Oracle("port").lib
        return Oracle._port

    # ------------------------------------------------------------ stencils
    def _unary(self, name, src, dst=None, *extra):
        src = np.ascontiguousarray(src, np.uint8)
        h, w = src.shape
        out = np.zeros_like(src) if dst is None else np.ascontiguousarray(dst, np.uint8).copy()
        if self.port:
            getattr(self.lib, "orc_" + name)(_p(out), _p(src), C.c_uint(w), C.c_uint(h), *extra)
        else:
            getattr(self.lib, "gs_" + name)(_img(out), _img(src), *extra)
        return out

    def blur(self, src, radius):
        return self._unary("blur", src, None, C.c_uint(radius))

    def sobel(self, src, dst=None):
        """dst: initial contents of the output (its 1-px frame is never written)."""
        return self._unary("sobel", src, dst)

    def erode(self, src):
        return self._unary("erode", src)

    def dilate(self, src):
        return self._unary("dilate", src)

    def adaptive_threshold(self, src, radius, c):
        return self._unary("adaptive_threshold", src, None, C.c_uint(radius), C.c_int(c))

    def filter(self, src, kernel, norm):
        src = np.ascontiguousarray(src, np.uint8)
        k = np.ascontiguousarray(kernel).astype(np.int8).view(np.uint8)
        h, w = src.shape
        kh, kw = k.shape
        out = np.zeros_like(src)
        if self.port:
            self.lib.orc_filter(_p(out), _p(src), C.c_uint(w), C.c_uint(h), _p(k), C.c_uint(kw),
                                C.c_uint(kh), C.c_uint(norm))
        else:
            self.lib.gs_filter(_img(out), _img(src), _img(k), C.c_uint(norm))
        return out

    def downsample(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        h, w = src.shape
        out = np.zeros((h // 2, w // 2), np.uint8)
        if self.port:
            self.lib.orc_downsample(_p(out), _p(src), C.c_uint(w), C.c_uint(h))
        else:
            self.lib.gs_downsample(_img(out), _img(src))
        return out

    # ---- geometry + template matching (ref :154-187, :705-739)
    def crop(self, src, rx, ry, rw, rh):
        src = np.ascontiguousarray(src, np.uint8)
        h, w = src.shape
        out = np.zeros((rh, rw), np.uint8)
        if self.port:
            self.lib.orc_crop(_p(out), C.c_uint(rw), C.c_uint(rh), _p(src), C.c_uint(w), C.c_uint(h),
                              C.c_uint(rx), C.c_uint(ry), C.c_uint(rw), C.c_uint(rh))
        else:
            self.lib.gs_crop.argtypes = [GsImage, GsImage, GsRect]
            self.lib.gs_crop(_img(out), _img(src), GsRect(rx, ry, rw, rh))
        return out

    def resize(self, src, dw, dh, nearest=False):
        src = np.ascontiguousarray(src, np.uint8)
        h, w = src.shape
        out = np.zeros((dh, dw), np.uint8)
        if self.port:
            f = self.lib.orc_resize_nn if nearest else self.lib.orc_resize
            f(_p(out), C.c_uint(dw), C.c_uint(dh), _p(src), C.c_uint(w), C.c_uint(h))
        else:
            f = self.lib.gs_resize_nn if nearest else self.lib.gs_resize
            f.argtypes = [GsImage, GsImage]
            f(_img(out), _img(src))
        return out

    def match_template(self, img, tmpl):
        img, tmpl = np.ascontiguousarray(img, np.uint8), np.ascontiguousarray(tmpl, np.uint8)
        (ih, iw), (th, tw) = img.shape, tmpl.shape
        out = np.zeros((ih - th + 1, iw - tw + 1), np.uint8)
        if self.port:
            self.lib.orc_match_template(_p(img), C.c_uint(iw), C.c_uint(ih), _p(tmpl), C.c_uint(tw), C.c_uint(th),
                                        _p(out))
        else:
            self.lib.gs_match_template.argtypes = [GsImage, GsImage, GsImage]
            self.lib.gs_match_template(_img(img), _img(tmpl), _img(out))
        return out

    def find_best_match(self, result):
        result = np.ascontiguousarray(result, np.uint8)
        h, w = result.shape
        if self.port:
            bx, by = C.c_uint(0), C.c_uint(0)
            self.lib.orc_find_best_match(_p(result), C.c_uint(w), C.c_uint(h), C.byref(bx), C.byref(by))
            return int(bx.value), int(by.value)
        self.lib.gs_find_best_match.argtypes = [GsImage]
        self.lib.gs_find_best_match.restype = GsPoint
        p = self.lib.gs_find_best_match(_img(result))
        return int(p.x), int(p.y)

    def histogram(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        hist = np.zeros(256, np.uint32)
        if self.port:
            self.lib.orc_histogram(_p(img), C.c_uint(w), C.c_uint(h), _p(hist))
        else:
            self.lib.gs_histogram(_img(img), _p(hist))
        return hist

    def otsu_threshold(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        if self.port:
            return int(self.lib.orc_otsu_threshold(_p(img), C.c_uint(w), C.c_uint(h)))
        return int(self.lib.gs_otsu_threshold(_img(img)))

    def otsu_from_hist(self, hist, npix):
        assert self.port
        hist = np.ascontiguousarray(hist, np.uint32)
        return int(self.lib.orc_otsu_from_hist(_p(hist), C.c_uint(npix)))

    def threshold(self, img, t):
        out = np.ascontiguousarray(img, np.uint8).copy()
        h, w = out.shape
        if self.port:
            self.lib.orc_threshold(_p(out), C.c_uint(w), C.c_uint(h), C.c_uint8(t))
        else:
            self.lib.gs_threshold(_img(out), C.c_uint8(t))
        return out

    # ------------------------------------------------------------ integral / LBP
    def integral(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        h, w = src.shape
        ii = np.zeros((h, w), np.uint32)
        if self.port:
            self.lib.orc_integral(_p(src), C.c_uint(w), C.c_uint(h), _p(ii))
        else:
            self.lib.gs_integral(_img(src), _p(ii))
        return ii

    def integral_sum(self, ii, x, y, w, h):
        ii = np.ascontiguousarray(ii, np.uint32)
        f = self.lib.orc_integral_sum if self.port else self.lib.ref_integral_sum
        f.restype = C.c_uint
        return int(f(_p(ii), C.c_uint(ii.shape[1]), C.c_uint(x), C.c_uint(y), C.c_uint(w), C.c_uint(h)))

    def lbp_window(self, cascade, ii, x, y, scale):
        ii = np.ascontiguousarray(ii, np.uint32)
        ih, iw = ii.shape
        f = self.lib.orc_lbp_window if self.port else self.lib.gs_lbp_window
        return int(f(C.addressof(cascade.as_struct()), _p(ii), iw, ih, x, y, scale))

    def lbp_detect(self, cascade, ii, max_rects, scale_factor, min_scale, max_scale, step):
        ii = np.ascontiguousarray(ii, np.uint32)
        ih, iw = ii.shape
        rects = np.zeros(max(max_rects, 1), RECT_DTYPE)
        f = self.lib.orc_lbp_detect if self.port else self.lib.gs_lbp_detect
        n = f(C.addressof(cascade.as_struct()), _p(ii), iw, ih, _p(rects), max_rects,
              scale_factor, min_scale, max_scale, step)
        return rects[:n].copy()

    # ------------------------------------------------------------ FAST / ORB
    def fast(self, img, nkps, threshold, scoremap=None):
        """returns (keypoints[n], scoremap).  scoremap: initial contents (3-px frame kept)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        sm = np.zeros_like(img) if scoremap is None else np.ascontiguousarray(scoremap).copy()
        kps = np.zeros(max(nkps, 1), KEYPOINT_DTYPE)
        if self.port:
            n = self.lib.orc_fast(_p(img), C.c_uint(w), C.c_uint(h), _p(sm), _p(kps),
                                  C.c_uint(nkps), C.c_uint(threshold))
        else:
            if w < 7 or h < 7:
                return kps[:0].copy(), sm  # the reference loops are empty (or wrap) here
            n = self.lib.gs_fast(_img(img), _img(sm), _p(kps), C.c_uint(nkps),
                                 C.c_uint(threshold))
        return kps[:n].copy(), sm

    def orientation(self, img, x, y, r=15):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        if self.port:
            return float(self.lib.orc_orientation(_p(img), C.c_uint(w), C.c_uint(h), C.c_uint(x),
                                                  C.c_uint(y), C.c_uint(r)))
        return float(self.lib.gs_compute_orientation(_img(img), x, y, r))

    def brief(self, img, x, y, angle):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        kp = np.zeros(1, KEYPOINT_DTYPE)
        kp["x"], kp["y"], kp["angle"] = x, y, angle
        if self.port:
            self.lib.orc_brief(_p(img), C.c_uint(w), C.c_uint(h), _p(kp))
        else:
            self.lib.gs_brief_descriptor(_img(img), _p(kp))
        return kp["desc"][0].copy()

    def orb_extract(self, img, nkps, threshold, scoremap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        sm = np.zeros_like(img) if scoremap is None else np.ascontiguousarray(scoremap).copy()
        kps = np.zeros(max(nkps, 1), KEYPOINT_DTYPE)
        if self.port:
            n = self.lib.orc_orb_extract(_p(img), C.c_uint(w), C.c_uint(h), _p(kps),
                                         C.c_uint(nkps), C.c_uint(threshold), _p(sm))
        else:
            if w < 7 or h < 7:
                return kps[:0].copy()
            n = self.lib.gs_orb_extract(_img(img), _p(kps), C.c_uint(nkps), C.c_uint(threshold),
                                        _p(sm))
        return kps[:n].copy()

    @staticmethod
    def orb_pyramid_buffer_bytes(w, h, n_levels):
        """levels 1.. back to back, then one scoremap per level (nanomagick.c:253-270)"""
        n_levels = min(n_levels, 4)
        dims = [(w, h)]
        for _ in range(1, n_levels):
            nw, nh = dims[-1][0] // 2, dims[-1][1] // 2
            if nw < 32 or nh < 32:
                break
            dims.append((nw, nh))
        return sum(a * b for a, b in dims[1:]) + sum(a * b for a, b in dims)

    def orb_extract_pyramid(self, img, nkps, threshold, n_levels, buffer=None):
        """ref examples/nanomagick/nanomagick.c:245-290; buffer = the caller's scratch bytes"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        nb = self.orb_pyramid_buffer_bytes(w, h, n_levels)
        buf = np.zeros(nb + 16, np.uint8) if buffer is None else np.ascontiguousarray(buffer, np.uint8).copy()
        assert buf.size >= nb
        kps = np.zeros(max(nkps, 1), KEYPOINT_DTYPE)
        if self.port:
            n = self.lib.orc_orb_extract_pyramid(_p(img), C.c_uint(w), C.c_uint(h), _p(kps), C.c_uint(nkps),
                                                 C.c_uint(threshold), _p(buf), C.c_uint(n_levels))
        else:
            nano = C.CDLL(os.path.join(HERE, "_ref", "libgs_ref_nano.so"))
            nano.ref_extract_pyramid_orb.restype = C.c_uint
            nano.ref_extract_pyramid_orb.argtypes = [GsImage, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_uint]
            n = nano.ref_extract_pyramid_orb(_img(img), _p(kps), nkps, threshold, _p(buf), n_levels)
        return kps[:n].copy(), buf

    def match_orb(self, k1, k2, max_matches, max_distance):
        k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        out = np.zeros(max(max_matches, 1), MATCH_DTYPE)
        f = self.lib.orc_match_orb if self.port else self.lib.gs_match_orb
        n = f(_p(k1), len(k1), _p(k2), len(k2), _p(out), max_matches, max_distance)
        return out[:n].copy()

    # libm as the reference calls it (grayskull.h:100-101)
    def atan2(self, y, x):
        assert not self.port
        return float(self.lib.ref_atan2(y, x))
