/*
 * k_lbp.h -- gs_lbp_detect / gs_lbp_window (grayskull.h:769-835) on gfx950.
 *
 * One thread = one detection window; a block covers one chunk (2048 consecutive windows of
 * ONE scale, row-major like the reference's y/x loops), so every wave evaluates the same weak
 * classifier of the same scale at the same time: feature geometry, leaf values and stage
 * thresholds are wave-uniform (scalar loads), only the 16 integral-image corners and the
 * 256-bit subset lookup are per-lane.  The integral image is read in the zero-bordered
 * (w+1) x (h+1) layout (k_integral_pad), so each of the nine 3x3-cell sums is
 * D + A - B - C on unguarded loads and the 36 loads of the reference collapse to the 16
 * distinct corners.  Stage sums are sequential float32 adds in weak order (ref :796-810).
 *
 * Not HBM-bound: the 8.3 MB table sits in L2/Infinity Cache; the bound is gather rate and
 * divergence (63 % of windows die in stage 0).  Survivors go through k_compact.h so the output
 * order is the reference's (scale, y, x) with the max_rects cap.
 */
#ifndef GS_K_LBP_H
#define GS_K_LBP_H
#include "k_compact.h"

namespace gs {

struct LbpScale {          /* one entry per visited scale (host-computed, float32 like ref :819-821) */
  int win_w, win_h;
  unsigned nx, ny;         /* window positions per row / column (step applied) */
  unsigned chunk_base;     /* first chunk of this scale in the frame's chunk array */
  unsigned nchunks;
};
struct LbpGeom { int off0, fw, fh_stride, pad; };      /* per (scale, weak): padded-table offsets */
struct LbpWeak { float left, right; unsigned sub_off, nsub; };
struct LbpStage { unsigned first, count; float threshold, pad; };

struct LbpArgs {
  const unsigned *padded;       /* n frames of (iw+1)*(ih+1) u32 */
  size_t frame_stride;          /* (iw+1)*(ih+1) */
  unsigned S;                   /* iw + 1 */
  size_t limit;                 /* frame_stride - 1 (GUARD clamp) */
  int step;
  unsigned nweaks, nstages;
  const LbpScale *scales;
  const LbpGeom *geom;          /* [nscales][nweaks] */
  const LbpWeak *weak;
  const LbpStage *stage;
  const int32_t *subsets;
  unsigned long long *mask;     /* n frames x total_chunks*kChunkWords */
  unsigned *chunk_count;        /* n frames x total_chunks */
  unsigned total_chunks;
};

template <bool GUARD>
GS_DEV bool lbp_window_pass(const LbpArgs &a, const unsigned *P, size_t origin,
                            const LbpGeom *geom) {
  /* GUARD: a scale whose (clamped, ref :803-804) feature rectangles can stick out of the
   * window reads clamped addresses instead of faulting; the reference reads out of bounds
   * there, so no particular value is "right". */
  auto ld = [&](long off) -> unsigned {
    size_t idx = origin + (size_t)off;
    if (GUARD) idx = idx > a.limit ? a.limit : idx;
    return P[idx];
  };
  for (unsigned s = 0; s < a.nstages; s++) {
    const LbpStage st = a.stage[s];
    float sum = 0.0f;
    for (unsigned k = 0; k < st.count; k++) {
      const unsigned wi = st.first + k;
      const LbpGeom g = geom[wi];
      const LbpWeak wk = a.weak[wi];
      unsigned G[4][4];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) G[j][i] = ld((long)g.off0 + (long)j * g.fh_stride + i * g.fw);
      unsigned c[3][3];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) c[j][i] = G[j + 1][i + 1] + G[j][i] - G[j][i + 1] - G[j + 1][i];
      const unsigned ctr = c[1][1];
      const unsigned code = ((c[0][0] >= ctr) << 7) | ((c[0][1] >= ctr) << 6) |
                            ((c[0][2] >= ctr) << 5) | ((c[1][2] >= ctr) << 4) |
                            ((c[2][2] >= ctr) << 3) | ((c[2][1] >= ctr) << 2) |
                            ((c[2][0] >= ctr) << 1) | ((c[1][0] >= ctr) << 0);
      const unsigned word = code >> 5, bit = code & 31u;
      bool hit = false;
      if (word < wk.nsub) hit = ((uint32_t)a.subsets[wk.sub_off + word] >> bit) & 1u;
      sum += hit ? wk.left : wk.right;
    }
    if (sum < st.threshold) return false;
  }
  return true;
}

/* grid (max chunks per scale, nscales, n frames), block 256: 8 windows per thread */
template <bool GUARD>
__global__ __launch_bounds__(256) void k_lbp_cascade(LbpArgs a) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  const LbpScale sc = a.scales[blockIdx.y];
  if (blockIdx.x >= sc.nchunks) return;
  const unsigned tid = threadIdx.x, wv = tid >> 6;
  const unsigned nwin = sc.nx * sc.ny;
  const unsigned *P = a.padded + (size_t)blockIdx.z * a.frame_stride;
  const LbpGeom *geom = a.geom + (size_t)blockIdx.y * a.nweaks;
  const size_t chunk = (size_t)blockIdx.z * a.total_chunks + sc.chunk_base + blockIdx.x;
  for (unsigned k = 0; k < kChunkItems / 256u; k++) {
    const unsigned idx = blockIdx.x * kChunkItems + k * 256u + tid;
    bool pass = false;
    if (idx < nwin) {
      const unsigned yi = idx / sc.nx, xi = idx - yi * sc.nx;
      const size_t origin = (size_t)(yi * (unsigned)a.step) * a.S + xi * (unsigned)a.step;
      pass = lbp_window_pass<GUARD>(a, P, origin, geom);
    }
    publish_flags(pass, a.mask, a.chunk_count, chunk * kChunkWords + k * 4u + wv);
  }
}

/* compaction functor: item -> gs_rect {x, y, win_w, win_h} (ref :825-828) */
struct LbpEmit {
  const LbpScale *scales;
  unsigned nscales;
  int step;
  unsigned *rects; /* n frames x max_rects x 4 u32 */
  unsigned max_rects;
  GS_DEV void operator()(unsigned frame, size_t item, unsigned r) const {
    const unsigned chunk = (unsigned)(item / kChunkItems);
    unsigned s = 0;
    while (s + 1 < nscales && scales[s + 1].chunk_base <= chunk) s++;
    const LbpScale sc = scales[s];
    const unsigned idx = (unsigned)(item - (size_t)sc.chunk_base * kChunkItems);
    const unsigned yi = idx / sc.nx, xi = idx - yi * sc.nx;
    unsigned *o = rects + ((size_t)frame * max_rects + r) * 4u;
    o[0] = xi * (unsigned)step, o[1] = yi * (unsigned)step;
    o[2] = (unsigned)sc.win_w, o[3] = (unsigned)sc.win_h;
  }
};

}  // namespace gs
#endif
