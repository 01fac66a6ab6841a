/*
 * lbp_types.h -- host-built tables of gs_lbp_detect that both the kernels (k_lbp.h) and the per-thread context
 * (gs_internal.h: the geometry cache) name.  Plain data, no kernels: any translation unit may include it.
 */
#ifndef GS_LBP_TYPES_H
#define GS_LBP_TYPES_H
namespace gs {
struct LbpScale {          /* one entry per visited scale (host-computed, float32 like ref :819-821) */
  int win_w, win_h;
  unsigned nx, ny;         /* window positions per row / column (step applied) */
  unsigned chunk_base;     /* first chunk of this scale in the frame's chunk array */
  unsigned nchunks;
};
struct LbpGeom { int off0, fw, fh_stride, pad; };      /* per (scale, weak): BYTE offsets in the padded table; pad = fh in rows */
}  // namespace gs
#endif
