"""LBP cascade container: the SoA tables of struct gs_lbp_cascade (grayskull.h:54-64).

`Cascade.from_blob` reads this repo's flat little-endian blob layout:
  "LBPC", u16 window_w, window_h, nfeatures, nweaks, nstages, pad, u32 nsubsets, then
  features i8[nf*4], weak_feature_idx u16[nw], left f32[nw], right f32[nw],
  subset_offset u16[nw], num_subsets u16[nw], subsets i32[nsub],
  stage_weak_start u16[ns], stage_nweaks u16[ns], stage_threshold f32[ns]
(each array padded to 4 bytes).  tests/golden/frontalface_cascade.bin is the
reference's frontalface cascade (examples/nanomagick/frontalface.h) in this layout.
"""
import ctypes as C
import struct

import numpy as np

from ._abi import GsLbpCascade

_FIELDS = [("features", np.int8), ("weak_feature_idx", np.uint16), ("weak_left_val", np.float32),
           ("weak_right_val", np.float32), ("weak_subset_offset", np.uint16),
           ("weak_num_subsets", np.uint16), ("subsets", np.int32),
           ("stage_weak_start", np.uint16), ("stage_nweaks", np.uint16),
           ("stage_threshold", np.float32)]


class Cascade:
    def __init__(self, window_w, window_h, **arrays):
        self.window_w, self.window_h = int(window_w), int(window_h)
        for name, dt in _FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays[name], dtype=dt))
        self.nfeatures = len(self.features) // 4
        self.nweaks = len(self.weak_feature_idx)
        self.nstages = len(self.stage_threshold)
        self._struct = None

    @classmethod
    def from_blob(cls, path):
        return cls.from_bytes(open(path, "rb").read(), path)

    @classmethod
    def from_bytes(cls, raw, origin="<bytes>"):
        """the blob as a byte string (what Sharder.broadcast_bytes hands every rank)"""
        if len(raw) < 20:
            raise ValueError("not a cascade blob: %r" % origin)
        magic, ww, wh, nf, nw, ns, _, nsub = struct.unpack_from("<4sHHHHHHI", raw, 0)
        if magic != b"LBPC":
            raise ValueError("not a cascade blob: %r" % origin)
        counts = dict(features=nf * 4, weak_feature_idx=nw, weak_left_val=nw, weak_right_val=nw,
                      weak_subset_offset=nw, weak_num_subsets=nw, subsets=nsub,
                      stage_weak_start=ns, stage_nweaks=ns, stage_threshold=ns)
        off, arrays = 20, {}
        for name, dt in _FIELDS:
            n = counts[name]
            if off + n * np.dtype(dt).itemsize > len(raw):
                raise ValueError("truncated cascade blob: %r" % origin)
            arrays[name] = np.frombuffer(raw, dtype=dt, count=n, offset=off).copy()
            off += (n * np.dtype(dt).itemsize + 3) & ~3
        return cls(ww, wh, **arrays)

    def as_struct(self):
        """ctypes struct gs_lbp_cascade pointing at this object's arrays (kept alive by self)."""
        if self._struct is None:
            s = GsLbpCascade(self.window_w, self.window_h, self.nfeatures, self.nweaks,
                             self.nstages)
            for name, _ in _FIELDS:
                setattr(s, name, getattr(self, name).ctypes.data)
            self._struct = s
        return self._struct

    def byref(self):
        return C.byref(self.as_struct())
