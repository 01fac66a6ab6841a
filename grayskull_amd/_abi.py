"""ctypes / numpy mirrors of the reference's C structs (grayskull.h:14-64).

Layouts are the x86-64 SysV ones the reference compiles to:
gs_image 16 B, gs_rect 16 B, gs_point 8 B, gs_keypoint 48 B, gs_match 12 B,
gs_lbp_cascade 96 B.  Checked by tests/test_abi.py against the C headers.
"""
import ctypes as C

import numpy as np


class GsImage(C.Structure):  # grayskull.h:14-17
    _fields_ = [("w", C.c_uint), ("h", C.c_uint), ("data", C.c_void_p)]


class GsRect(C.Structure):  # grayskull.h:19-21
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("w", C.c_uint), ("h", C.c_uint)]


class GsPoint(C.Structure):  # grayskull.h:23-25
    _fields_ = [("x", C.c_uint), ("y", C.c_uint)]


class GsKeypoint(C.Structure):  # grayskull.h:42-47
    _fields_ = [("pt", GsPoint), ("response", C.c_uint), ("angle", C.c_float),
                ("descriptor", C.c_uint32 * 8)]


class GsMatch(C.Structure):  # grayskull.h:49-52
    _fields_ = [("idx1", C.c_uint), ("idx2", C.c_uint), ("distance", C.c_uint)]


class GsLbpCascade(C.Structure):  # grayskull.h:54-64
    _fields_ = [("window_w", C.c_uint16), ("window_h", C.c_uint16),
                ("nfeatures", C.c_uint16), ("nweaks", C.c_uint16), ("nstages", C.c_uint16),
                ("features", C.c_void_p), ("weak_feature_idx", C.c_void_p),
                ("weak_left_val", C.c_void_p), ("weak_right_val", C.c_void_p),
                ("weak_subset_offset", C.c_void_p), ("weak_num_subsets", C.c_void_p),
                ("subsets", C.c_void_p), ("stage_weak_start", C.c_void_p),
                ("stage_nweaks", C.c_void_p), ("stage_threshold", C.c_void_p)]


KEYPOINT_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("response", "<u4"), ("angle", "<f4"),
                           ("desc", "<u4", (8,))])
RECT_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("w", "<u4"), ("h", "<u4")])
MATCH_DTYPE = np.dtype([("idx1", "<u4"), ("idx2", "<u4"), ("distance", "<u4")])

assert C.sizeof(GsImage) == 16 and C.sizeof(GsRect) == 16 and C.sizeof(GsPoint) == 8
assert C.sizeof(GsKeypoint) == 48 == KEYPOINT_DTYPE.itemsize
assert C.sizeof(GsMatch) == 12 == MATCH_DTYPE.itemsize
assert C.sizeof(GsLbpCascade) == 96 and RECT_DTYPE.itemsize == 16
