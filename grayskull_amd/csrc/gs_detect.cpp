/*
 * gs_detect.cpp -- launchers and C ABI of the detectors of libgrayskull_hip.so: the ordered, capped compaction, gs_fast,
 * the LBP cascade (gs_lbp_detect / gs_lbp_window), gs_orb_extract and its halves, gs_match_orb.  The only arithmetic done
 * on the host is what the reference itself delegates to libm (atan2f / sinf, grayskull.h:100-101), the float32 scale
 * progression of gs_lbp_detect (ref :819-821, :799-804) and the stable sort of <= 5000 candidates (ref :639).
 */
#include "gs_internal.h"
#include <pthread.h>

#include "k_fast.h"
#include "k_fast_nms.h"
#include "k_lbp.h"
#include "k_lbp_tile.h"
#include "k_orb.h"

namespace gsi {
thread_local unsigned long long *g_lbp_evaluated = nullptr;
}

namespace {

/* ------------------------------------------------------------------ ordered compaction driver */
template <bool QUAD = false, class F>
void run_compaction(unsigned long long *mask, unsigned *cnt, unsigned nchunks, unsigned n,
                    unsigned cap, unsigned *totals_dev, F emit, hipStream_t on = nullptr, unsigned *pfx_in = nullptr) {
  hipStream_t st = on ? on : ctx().s();
  if (nchunks <= kEmitSelfScan) { /* FAST on video frames, match: one launch less on a latency-bound tail */
    GS_LAUNCH((k_emit<F, QUAD>), dim3((nchunks + 3) / 4, n), dim3(256), 0, st,
              (const unsigned long long *)mask, (const unsigned *)cnt, (const unsigned *)nullptr, nchunks,
              cap, emit, totals_dev);
    return;
  }
  unsigned *pfx = pfx_in ? pfx_in : (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
  GS_LAUNCH(k_chunk_scan, dim3(n), dim3(1024), 0, st, (const unsigned *)cnt, nchunks, pfx,
            totals_dev, cap);
  GS_LAUNCH((k_emit<F, QUAD>), dim3((nchunks + 3) / 4, n), dim3(256), 0, st,
            (const unsigned long long *)mask, (const unsigned *)cnt, (const unsigned *)pfx, nchunks,
            cap, emit, (unsigned *)nullptr);
}

/* ------------------------------------------------------------------ FAST */
/* gs_fast pass 1 (w, h >= 7, n <= kMaxZ): k_fast_score_q4 (LDS tile, 4 px per thread through the compass filter, candidates
 * queued; 48-row tiles: 32 x 720p block noise 101 us at 16 rows, 94 at 32, 92 at 48 / 64, profiles/r04l_fast_tile_rows.log),
 * which also leaves the bitmap of scored pixels for the sparse NMS pass.  Thresholds above 0xffffff00 (p + t wraps in 32 bits)
 * and gsh_tune key 7 = 2 take k_fast_score_px: one global byte load per ring pixel, the literal form of ref :491-513.
 * Returns whether the bitmap was written. */
bool fast_score_leaves_bitmap(unsigned threshold) { return g_tune[7] != 2 && threshold <= 0xffffff00u; }
bool launch_fast_score(hipStream_t on, const uint8_t *img, uint8_t *score, unsigned w, unsigned h, unsigned n,
                       unsigned threshold, unsigned long long *nz = nullptr, size_t nz_frame_words = 0) {
  const size_t fb = (size_t)w * h;
  if (!fast_score_leaves_bitmap(threshold)) {
    GS_LAUNCH(k_fast_score_px, grid2d(w - 6, h - 6, n), dim3(64, 4), 0, on, img, score, w, h, fb, threshold);
    return false;
  }
  constexpr unsigned rows = 48;
  const unsigned tx = (w - 6 + 63) / 64, ty = (h - 6 + rows - 1) / rows;
  const unsigned long long nt = (unsigned long long)tx * ty * n;
  GS_ASSERT(nt < (1ull << 24)); /* the kernel's 32-bit strides (tile index) stay clear of 2^32: 2^24 tiles = 2^34 pixels per call */
  const unsigned share = (g_tune[18] == 1 || !topo().eight_xcds()) ? 0u : (unsigned)((nt + 7) / 8); /* key 18 = 1: tiles in launch order */
  const dim3 grid(share ? share * 8u : (unsigned)nt), block(64, 4);
  const auto magic = [](unsigned d) { return d <= 1u ? 0xffffffffu : (unsigned)((1ull << 32) / d); }; /* floor(2^32 / d) */
  GS_LAUNCH(k_fast_score_q4<rows>, grid, block, 0, on, img, score, w, h, fb, threshold, tx, ty, (unsigned)nt, share, nz, nz_frame_words,
            magic(tx), magic(tx * ty));
  return true;
}

/* clip_w / clip_h (single frame only): the caller's score map is smaller than the image; positions
 * outside it read 0 in the NMS pass like gs_get does (ref :524) */
void launch_fast(const uint8_t *img, uint8_t *score, unsigned w, unsigned h, unsigned n,
                 unsigned *kps, unsigned *counts, unsigned nkps, unsigned threshold, unsigned clip_w = 0,
                 unsigned clip_h = 0) {
  hipStream_t st = ctx().s();
  if (n == 0) return;
  if (n > kMaxZ) { /* grid.y / grid.z carry the frame index: split like every other launcher */
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ)
      launch_fast(img + (size_t)w * h * f0, score + (size_t)w * h * f0, w, h, std::min(kMaxZ, n - f0),
                  kps + (size_t)f0 * nkps * 12, counts + f0, nkps, threshold);
    return;
  }
  if (w < 7 || h < 7) { /* reference loops are empty for 3 <= dim < 7 */
    GS_HIP(hipMemsetAsync(counts, 0, (size_t)n * 4, st));
    return;
  }
  const size_t fb = (size_t)w * h;
  /* (Both passes in one walk -- score tile, NMS and mask words from LDS, round 3's k_fast_fused -- measured 127 us against
   * 63 + 23 per 32 x 720p and was removed in round 4: scripts/experiments/not_kept/, profiles/r03t_fast_fused_not_kept.log.) */
  /* pass 2, sparse (k_fast_nms.h): the score kernel leaves a bitmap of the scored pixels, one word per 64-px tile row of
   * the interior, and only those are tested.  Key 19 = 1, and the calls whose score pass is k_fast_score_px: the
   * item-by-item kernel k_fast_nms below. */
  {
    const unsigned tx = (w - 6 + 63) / 64;
    const unsigned long long nw = (unsigned long long)tx * (h - 6);
    if (g_tune[19] != 1 && fast_score_leaves_bitmap(threshold) && nw * 64 < (1ull << 32)) {
      const unsigned nwords = (unsigned)nw, nchunks = (nwords + kChunkWords - 1) / kChunkWords;
      GS_ASSERT((unsigned long long)n * nchunks < (1ull << 32));
      unsigned long long *nz = (unsigned long long *)ctx().scratch(SL_NZ, (size_t)n * nchunks * kChunkWords * 8);
      unsigned long long *mask = (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nchunks * kChunkWords * 8);
      unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)n * nchunks * 4);
      unsigned *pfx = (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
      launch_fast_score(st, img, score, w, h, n, threshold, nz, (size_t)nchunks * kChunkWords);
      if (clip_w && n == 1 && (clip_w < w || clip_h < h))
        GS_LAUNCH(k_fast_clip, grid2d(w, h, 1), dim3(64, 4), 0, st, score, w, h, clip_w, clip_h);
      GS_LAUNCH(k_fast_nms_sparse, dim3((nchunks + 3) / 4, n), dim3(256), 0, st, (const uint8_t *)score, w, fb,
                (const unsigned long long *)nz, mask, cnt, tx, nwords, nchunks);
      run_compaction(mask, cnt, nchunks, n, nkps, counts,
                     FastEmitPadded{score, w, tx * 64u, fb, kps, nkps, ((uintptr_t)kps & 15) == 0, 3u}, st, pfx);
      return;
    }
  }
  launch_fast_score(st, img, score, w, h, n, threshold);
  const unsigned nitems = (w - 6) * (h - 6);
  const unsigned nchunks = (nitems + kChunkItems - 1) / kChunkItems;
  unsigned long long *mask =
      (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nchunks * kChunkWords * 8);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)n * nchunks * 4);
  unsigned *pfx = (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
  /* row = item / (w-6) by multiplication where the magic fits (see div_by) */
  const unsigned iw = w - 6;
  const unsigned magic = (iw > 256 && iw <= 8192 && (unsigned long long)iw * (h - 6) <= (1ull << 26))
                             ? (unsigned)(((1ull << 40) + iw - 1) / iw) : 0u;
  /* NMS flags, chunk scan, ordered emit of frames [f0, f0 + nn) on stream `on` */
  auto rest = [&](hipStream_t on, unsigned f0, unsigned nn) {
    if (clip_w && n == 1 && (clip_w < w || clip_h < h))
      GS_LAUNCH(k_fast_clip, grid2d(w, h, 1), dim3(64, 4), 0, on, score, w, h, clip_w, clip_h);
    unsigned *c = cnt + (size_t)f0 * nchunks; /* k_fast_nms stores every chunk's count: no zeroing */
    GS_LAUNCH(k_fast_nms, dim3(nchunks, nn), dim3(256), 0, on, (const uint8_t *)score + fb * f0, w, h, fb,
              mask + (size_t)f0 * nchunks * kChunkWords, c, nchunks, magic);
    run_compaction</*QUAD=*/true>(mask + (size_t)f0 * nchunks * kChunkWords, c, nchunks, nn, nkps, counts + f0, /* k_fast_nms: 4 items per lane */
                                  FastEmit{score + fb * f0, w, fb, kps + (size_t)f0 * nkps * 12, nkps, ((uintptr_t)kps & 15) == 0}, on,
                                  pfx + (size_t)f0 * nchunks);
  };
  /* Tried and not kept: cutting a batch into 2-8 groups of frames and running group i's NMS / scan / emit on the side
   * stream under group i+1's score pass.  The cross-stream event hops cost more than the ~60 us of small passes
   * they could hide: 32 x 720p 4.2 us per frame in one piece, 5.1 in two, 6.8 in four (profiles/r02i_fast_groups_not_kept.log). */
  rest(st, 0, n);
}

/* ------------------------------------------------------------------ LBP cascade */
}  // namespace

struct gsh_cascade {
  unsigned window_w, window_h, nfeatures, nweaks, nstages, nsub = 0;
  std::vector<int8_t> features;
  std::vector<uint16_t> weak_feature_idx;
  LbpWeak *d_weak = nullptr;
  LbpStage *d_stage = nullptr;
  int32_t *d_subsets = nullptr;
  uint32_t *d_truth = nullptr; /* per-stage truth tables (LbpStage::truth) */
  unsigned ntruth = 0;
  unsigned long long id = 0; /* unique per handle: key of the calling threads' geometry caches */
};

namespace gsi {
/* the device tables of a handle, without touching any context (used while a context is being released) */
void gsh_cascade_tables_deleter::operator()(gsh_cascade *dc) const {
  (void)hipFree(dc->d_weak), (void)hipFree(dc->d_stage), (void)hipFree(dc->d_subsets), (void)hipFree(dc->d_truth);
  delete dc;
}
}  // namespace gsi

namespace {

/* The reference's scale loop and per-feature truncation (ref :819-821, :799-804), float32. */
void build_scales(const gsh_cascade &c, unsigned iw, unsigned ih, float scale_factor,
                  float min_scale, float max_scale, int step, std::vector<LbpScale> &scales,
                  std::vector<LbpGeom> &geom, bool &guard, unsigned long long &nwin) {
  scales.clear();
  geom.clear();
  guard = false;
  nwin = 0;
  unsigned chunk_base = 0;
  const unsigned S = iw + 1;
  for (float scale = min_scale; scale <= max_scale; scale *= scale_factor) {
    const int win_w = (int)((int)c.window_w * scale), win_h = (int)((int)c.window_h * scale);
    if (win_w > (int)iw || win_h > (int)ih) break;
    LbpScale sc;
    sc.win_w = win_w, sc.win_h = win_h;
    sc.nx = ((unsigned)((int)iw - win_w)) / (unsigned)step + 1;
    sc.ny = ((unsigned)((int)ih - win_h)) / (unsigned)step + 1;
    sc.chunk_base = chunk_base;
    sc.nchunks = (sc.nx * sc.ny + kChunkItems - 1) / kChunkItems;
    chunk_base += sc.nchunks;
    nwin += (unsigned long long)sc.nx * sc.ny;
    for (unsigned wi = 0; wi < c.nweaks; wi++) {
      const int fi = c.weak_feature_idx[wi];
      int fx = (int)((int)c.features[fi * 4 + 0] * scale);
      int fy = (int)((int)c.features[fi * 4 + 1] * scale);
      int fw = (int)((int)c.features[fi * 4 + 2] * scale);
      int fh = (int)((int)c.features[fi * 4 + 3] * scale);
      if (fw < 1) fw = 1;
      if (fh < 1) fh = 1;
      if (fx < 0 || fy < 0 || fx + 3 * fw > win_w || fy + 3 * fh > win_h) guard = true;
      geom.push_back(LbpGeom{(fy * (int)S + fx) * 4, fw * 4, fh * (int)S * 4, fh});
    }
    scales.push_back(sc);
    if (scales.size() >= 4096 || !(scale_factor > 1.0f)) break; /* the reference would not terminate */
  }
}

LbpGeomCache &cascade_prepare(const gsh_cascade *dc, unsigned iw, unsigned ih, float sf, float mn, float mx,
                              int step) {
  Ctx &cx = ctx();
  if (cx.geom_cache.size() > 16 && !cx.geom_cache.count(dc->id)) { /* handles come and go: bound the cache */
    cx.sync();
    cx.drop_geom();
  }
  LbpGeomCache &gc = cx.geom_cache[dc->id];
  if (gc.iw == iw && gc.ih == ih && gc.sf == sf && gc.mn == mn && gc.mx == mx && gc.step == step && gc.d_scales)
    return gc;
  std::vector<LbpGeom> geom;
  build_scales(*dc, iw, ih, sf, mn, mx, step, gc.scales, geom, gc.guard, gc.nwindows);
  cx.sync(); /* tables may be in use by an earlier launch of this thread */
  const size_t sb = std::max<size_t>(1, gc.scales.size()) * sizeof(LbpScale);
  const size_t gb = std::max<size_t>(1, geom.size()) * sizeof(LbpGeom);
  if (gc.d_scales_cap < sb) {
    if (gc.d_scales) GS_HIP(hipFree(gc.d_scales));
    GS_HIP(hipMalloc((void **)&gc.d_scales, sb));
    gc.d_scales_cap = sb;
  }
  if (gc.d_geom_cap < gb) {
    if (gc.d_geom) GS_HIP(hipFree(gc.d_geom));
    GS_HIP(hipMalloc((void **)&gc.d_geom, gb));
    gc.d_geom_cap = gb;
  }
  if (!gc.scales.empty()) {
    GS_HIP(hipMemcpy(gc.d_scales, gc.scales.data(), gc.scales.size() * sizeof(LbpScale), hipMemcpyHostToDevice));
    GS_HIP(hipMemcpy(gc.d_geom, geom.data(), geom.size() * sizeof(LbpGeom), hipMemcpyHostToDevice));
  }
  gc.total_chunks = 0, gc.max_chunks = 0;
  for (auto &sc : gc.scales) {
    gc.total_chunks += sc.nchunks;
    gc.max_chunks = std::max(gc.max_chunks, sc.nchunks);
  }
  gc.iw = iw, gc.ih = ih, gc.sf = sf, gc.mn = mn, gc.mx = mx, gc.step = step;
  return gc;
}

/* padded: n frames of (iw+1)*(ih+1) u32 on device */
void launch_lbp_padded(const gsh_cascade *dc, const LbpGeomCache &gc, const unsigned *padded, unsigned iw,
                       unsigned ih, unsigned n, unsigned *rects, unsigned *counts, unsigned max_rects,
                       int step) {
  hipStream_t st = ctx().s();
  if (gc.scales.empty() || max_rects == 0) {
    GS_HIP(hipMemsetAsync(counts, 0, (size_t)n * 4, st));
    return;
  }
  const unsigned nch = gc.total_chunks;
  unsigned long long *mask =
      (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nch * kChunkWords * 8);
  const unsigned nsc0 = (unsigned)gc.scales.size();
  /* chunk counters, then the early-exit counters: one per group of 32 chunks, one per 1024 */
  const unsigned ngroups = (nch >> kLbpGroupShift) + 1, nsupers = (nch >> kLbpSuperShift) + 1;
  const size_t ncnt = (size_t)n * nch + (size_t)n * ngroups + (size_t)n * nsupers + n;
  unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, ncnt * 4);
  GS_HIP(hipMemsetAsync(cnt, 0, ncnt * 4, st));
  GS_HIP(hipMemsetAsync(mask, 0, (size_t)n * nch * kChunkWords * 8, st));
  LbpArgs a;
  a.hits_group = cnt + (size_t)n * nch;
  a.hits_super = a.hits_group + (size_t)n * ngroups;
  a.hits_total = a.hits_super + (size_t)n * nsupers;
  a.scale0 = 0;
  a.ngroups = ngroups, a.nsupers = nsupers;
  a.evaluated = g_lbp_evaluated;
  a.nscales = nsc0, a.cap = max_rects;
  a.nwindows_cap = (unsigned)std::min<unsigned long long>(gc.nwindows, 0xffffffffull);
  a.padded = padded;
  a.frame_stride = (size_t)(iw + 1) * (ih + 1);
  a.S = iw + 1;
  a.limit_bytes = (unsigned)((a.frame_stride - 1) * 4);
  a.step = step;
  a.nweaks = dc->nweaks, a.nstages = dc->nstages, a.nsub = dc->nsub;
  a.scales = gc.d_scales, a.geom = gc.d_geom, a.weak = dc->d_weak, a.stage = dc->d_stage;
  a.subsets = dc->d_subsets;
  a.truth = dc->d_truth, a.ntruth = dc->ntruth;
  a.mask = mask, a.chunk_count = cnt, a.total_chunks = nch;
  const unsigned nsc = (unsigned)gc.scales.size();
  /* XCD-aware chunk mapping (k_lbp.h) once the integral image no longer fits one XCD's 4 MB L2: 1080p -3 %, 4K block
   * noise -4 %, 4K edge maps -12 % (5.76 -> 5.08 ms per frame); 720p (3.7 MB) is 1-4 % better off in dispatch order
   * (profiles/r02l_lbp_xcd.log).  Key 13: 1 = never, 2 = always. */
  a.xcd_swizzle = g_tune[13] == 1 ? 0u : g_tune[13] >= 2 ? 1u : ((a.frame_stride * 4 >= (size_t)6 << 20 && topo().eight_xcds()) ? 1u : 0u);
  /* k_lbp_tile's tiles go out in dispatch order = the reference's scan order within a scale, whatever the table's size: a
   * tile is read once into the LDS, so the L2 mapping is worth 1 % (eighths: 1080p block noise 0.621 vs 0.628 ms), while a
   * frame that reaches max_rects stops evaluating sooner when the tiles before the cap run first -- configs[4]'s 4K edge maps
   * (4096 rectangles reached in the last scale) 3.41 -> 3.19 ms per frame; per scale, uncapped, the two mappings are equal
   * (profiles/r05j_lbp_xcd*.log).  Key 13 = 2: eighths for the tiles too; 16 + G: runs of G tiles dealt round the XCDs
   * (not kept: 2 and 4 equal dispatch order, 8 and 16 are 1-10 % slower). */
  const unsigned tile_map = g_tune[13] >= 17 ? (unsigned)g_tune[13] : g_tune[13] == 2 ? 1u : 0u;
  const size_t lds = (size_t)dc->nstages * sizeof(LbpStage) +
                     (size_t)dc->nweaks * (sizeof(LbpWeak) + sizeof(LbpGeom)) + (size_t)dc->nsub * 4;
  GS_ASSERT(lds <= 60 * 1024 && "cascade tables must fit the block's LDS");
  /* phases of the block-local survivor re-packing: g_tune[4] selects a preset */
  LbpPhases ph;
  {
    static const unsigned presets[8][kLbpMaxPhases] = {
        {2, 4, 7, 99, 99, 99, 99, 99},    /* 0 default (measured best on MI355X, frontalface) */
        {99, 99, 99, 99, 99, 99, 99, 99}, /* 1: single dense phase (no re-packing) */
        {1, 2, 4, 6, 9, 13, 99, 99},
        {1, 2, 3, 4, 6, 8, 12, 99},
        {1, 2, 5, 99, 99, 99, 99, 99},
        {2, 5, 99, 99, 99, 99, 99, 99},
        {2, 6, 99, 99, 99, 99, 99, 99},
        {3, 7, 99, 99, 99, 99, 99, 99}};
    unsigned custom[kLbpMaxPhases];
    const unsigned *pr = presets[(g_tune[4] >= 0 && g_tune[4] < 8) ? g_tune[4] : 0];
    if (g_tune[4] >= 1000) { /* experiments: 1000 + e0 + 32 e1 + 1024 e2 + 32768 e3 (0 = no further split) */
      unsigned v = (unsigned)g_tune[4] - 1000u;
      for (unsigned i = 0; i < kLbpMaxPhases; i++, v >>= 5) custom[i] = (v & 31u) ? (v & 31u) : 99u;
      pr = custom;
    }
    ph.n = 0;
    unsigned prev = 0;
    for (unsigned i = 0; i < kLbpMaxPhases && prev < dc->nstages; i++) {
      const unsigned e = (i + 1 == kLbpMaxPhases) ? dc->nstages : std::min(pr[i], dc->nstages);
      if (e <= prev) continue;
      ph.end[ph.n++] = e, prev = e;
    }
    if (ph.n == 0) ph.n = 1, ph.end[0] = dc->nstages;
    ph.end[ph.n - 1] = dc->nstages;
    /* preset 0 (default): first re-packing point chosen per block between stages 2 and 8 (k_lbp.h) */
    ph.adaptive_max = (g_tune[4] == 0 && dc->nstages > 2) ? 8u : 0u; /* 6 .. 15 within 1.5 % (profiles/r02l_lbp_adaptive_xcd.log) */
    ph.adaptive_tenths = 2u;
    ph.quad = g_tune[17] == 1 ? 0u : 1u; /* key 17 = 1: one lane per re-packed window (the round-2 form) */
    /* with quad-lane survivors (profiles/r03f_lbp_adaptive_quad.log, r03k_lbp_adaptive_next.log): +1 +2 +4 is best on
     * block noise (8 x 1080p 0.76 vs 0.80 ms for +1 +3 +6, 4K 3.15 vs 3.17) and within 0.5 % of the best on edge maps */
    ph.adaptive_next[0] = 1u, ph.adaptive_next[1] = 2u, ph.adaptive_next[2] = 4u;
    ph.tile_first = 1u, ph.tile_tenths = 7u;
    if (g_tune[15] > 0) { /* experiments: first + 16 * tenths.  first = 0 (e.g. key 15 = 112) means "no dense stage": the dense loop still
                           * has to run stage 0 for a wave to have survivors at all, so it is taken as 1 (results never depend on a key) */
      ph.tile_first = std::max(1u, (unsigned)g_tune[15] & 15u), ph.tile_tenths = ((unsigned)g_tune[15] >> 4) & 15u;
    }
    GS_ASSERT(ph.tile_first >= 1u && ph.tile_first <= 15u && ph.tile_tenths <= 15u);
    if (ph.adaptive_max && g_tune[9] > 0) { /* experiments: key 9 = max + 16 * tenths (+ 256 d1 + 4096 d2 + 65536 d3: later points) */
      const unsigned v = (unsigned)g_tune[9];
      ph.adaptive_max = v & 15u, ph.adaptive_tenths = (v >> 4) & 15u;
      if (v >> 8) ph.adaptive_next[0] = (v >> 8) & 15u, ph.adaptive_next[1] = (v >> 12) & 15u, ph.adaptive_next[2] = (v >> 16) & 15u;
    }
  }
  const size_t lds_all = ((lds + 15) & ~(size_t)15) + 2 * kChunkItems * 2 + 64 * 4 + 16;
  /* (Round 3's row-sharing stage prefilter k_lbp_dense -- stages 0..1 for every window with the table rows handed down
   * the columns of 64 x 64-window tiles, 2.8x fewer gathers per classifier -- lost to the dense phase it replaced: on the
   * configs[4] input 56.9 + 21.6 ms against 71.2 ms per 16 frames, half of its L2 requests missing and 62 % of its wave
   * cycles stalled (profiles/r04j_lbp_counters_*.txt).  Removed in round 4; sources in scripts/experiments/not_kept/.)
   *
   * Round 5: scales whose table tile fits the LDS run on k_lbp_tile (k_lbp_tile.h: corners from an LDS tile, survivors one
   * lane per (window, classifier) pair); the others -- and GUARD geometries, fixed phase presets -- on k_lbp_cascade.
   * Consecutive scales with the same choice share a launch; launches go out in scale order, which the max_rects exit
   * wants anyway.  Key 14: 0 = by rule, 1 = k_lbp_cascade for every scale, 2 + i = tile shape i wherever it fits. */
  struct TileCfg {
    unsigned nt, tw, th, want_blocks; /* want_blocks: the rule takes the shape when at least that many blocks fit a CU */
    void (*fn)(LbpArgs, LbpPhases);
    void (*fn_count)(LbpArgs, LbpPhases);
  };
  /* in the rule's order of preference (profiles/r05a_lbp_tile.log, r05b_*: per-scale times of every shape) */
  static const TileCfg cfgs[] = {
      {512, 128, 32, 3, k_lbp_tile<512, 128, 32>, k_lbp_tile<512, 128, 32, true>},
      {1024, 128, 32, 2, k_lbp_tile<1024, 128, 32>, k_lbp_tile<1024, 128, 32, true>},
      {1024, 64, 32, 2, k_lbp_tile<1024, 64, 32>, k_lbp_tile<1024, 64, 32, true>},
      {1024, 64, 16, 2, k_lbp_tile<1024, 64, 16>, k_lbp_tile<1024, 64, 16, true>},
      {512, 64, 32, 2, k_lbp_tile<512, 64, 32>, k_lbp_tile<512, 64, 32, true>},
  };
  constexpr int kNumCfgs = (int)(sizeof(cfgs) / sizeof(cfgs[0]));
  constexpr size_t kLdsPerCu = 160 * 1024, kLdsDyn = kLdsPerCu - 1024; /* the kernels' static __shared__ words count against the CU's LDS */
#ifndef GS_EMU
  if (!ctx().lbp_lds_raised) { /* per thread + device, cleared by Ctx::release(): a re-initialised runtime gets its attributes again */
    for (const TileCfg &c : cfgs) {
      GS_HIP(hipFuncSetAttribute((const void *)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsDyn));
      GS_HIP(hipFuncSetAttribute((const void *)c.fn_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsDyn));
    }
    ctx().lbp_lds_raised = true;
  }
#endif
  auto tile_lds = [&](const TileCfg &c, const LbpScale &sc) {
    const size_t ts = lbp_tile_stride(c.tw, (unsigned)step, (unsigned)sc.win_w), tr = (size_t)(c.th - 1) * step + sc.win_h + 1;
    return lbp_tile_lds_bytes(dc->nstages, dc->nweaks, dc->nsub, dc->ntruth, c.tw * c.th, ts * tr);
  };
  /* blocks a CU holds: LDS (a block's dynamic part + the kernel's few static words, in the allocator's 512-byte granules --
   * counting a whole KB per block put scale 2.36 of a 4K scan at one block per CU when two fit: 0.30 instead of 0.23 ms) and
   * 32 waves */
  auto tile_blocks = [&](const TileCfg &c, const LbpScale &sc) {
    const size_t need = tile_lds(c, sc);
    return need > kLdsDyn ? 0u : (unsigned)std::min<size_t>(kLdsPerCu / ((need + 64 + 511) & ~(size_t)511), 32u * 64u / c.nt);
  };
  struct Choice { int cfg; unsigned blocks; bool operator==(const Choice &o) const { return cfg == o.cfg && blocks == o.blocks; } };
  auto tile_choice = [&](const LbpScale &sc) -> Choice { /* cfg -1: k_lbp_cascade */
    if (gc.guard || !ph.adaptive_max || g_tune[14] == 1) return Choice{-1, 0};
    if (g_tune[14] >= 2) { /* experiments: one shape wherever its tile fits */
      const int i = g_tune[14] - 2;
      const unsigned b = i < kNumCfgs ? tile_blocks(cfgs[i], sc) : 0u;
      return b ? Choice{i, b} : Choice{-1, 0};
    }
    /* profiles/r05c_lbp_tile_v3_own_tables_addc.log, per scale: 512 threads on 128 x 32 windows while three blocks fit a CU
     * (scales 1, 1.1), 1024 threads on it while two fit (to 2.36), then 1024 threads on 64 x 32 while two fit (to 3.45), and
     * for what is left (3.8: 124 KB of tile) one 1024-thread block per CU still beats k_lbp_cascade (0.47 vs 0.52 ms); the
     * last two shapes lost at every scale */
    for (int i = 0; i < 3; i++) {
      const unsigned b = tile_blocks(cfgs[i], sc);
      if (b >= cfgs[i].want_blocks) return Choice{i, b};
    }
    if (g_tune[14] != -1 && tile_blocks(cfgs[1], sc) >= 1u) return Choice{1, 1u}; /* key 14 = -1: k_lbp_cascade for these */
    return Choice{-1, 0};
  };
  /* Consecutive scales with the same shape AND the same blocks per CU share a launch (its dynamic LDS is the largest
   * member's, which by construction still fits that many blocks: a launch never lowers a member's occupancy -- with one
   * launch for all scales the small ones ran at the big ones' one block per CU, 14 instead of 4 ms). */
  for (unsigned s0 = 0; s0 < nsc;) {
    const Choice choice = tile_choice(gc.scales[s0]);
    unsigned s1 = s0 + 1;
    while (s1 < nsc && tile_choice(gc.scales[s1]) == choice) s1++;
    a.scale0 = s0;
    if (choice.cfg < 0) {
      unsigned mc = 0;
      for (unsigned s = s0; s < s1; s++) mc = std::max(mc, gc.scales[s].nchunks);
      const dim3 g(a.xcd_swizzle ? (mc + 7u) & ~7u : mc, s1 - s0, n);
      if (a.evaluated) { /* counting build: the same kernel + one register that counts classifier evaluations */
        if (gc.guard) GS_LAUNCH((k_lbp_cascade<true, true>), g, dim3(256), lds_all, st, a, ph);
        else GS_LAUNCH((k_lbp_cascade<false, true>), g, dim3(256), lds_all, st, a, ph);
      } else if (gc.guard) GS_LAUNCH(k_lbp_cascade<true>, g, dim3(256), lds_all, st, a, ph);
      else GS_LAUNCH(k_lbp_cascade<false>, g, dim3(256), lds_all, st, a, ph);
    } else {
      const TileCfg &c = cfgs[choice.cfg];
      unsigned mt = 0;
      size_t need = 0;
      LbpArgs at = a;
      at.xcd_swizzle = tile_map;
      for (unsigned s = s0; s < s1; s++) {
        const LbpScale &sc = gc.scales[s];
        const unsigned tx = (sc.nx + c.tw - 1) / c.tw, ty = (sc.ny + c.th - 1) / c.th;
        const unsigned per = tile_map >= 16u ? 8u * (tile_map - 16u) : tile_map == 1u ? 8u : 1u;
        mt = std::max(mt, (tx * ty + per - 1u) / per * per);
        need = std::max(need, tile_lds(c, sc));
      }
      const dim3 g(mt, s1 - s0, n);
      if (a.evaluated) GS_LAUNCH(c.fn_count, g, dim3(c.nt), need, st, at, ph);
      else GS_LAUNCH(c.fn, g, dim3(c.nt), need, st, at, ph);
    }
    s0 = s1;
  }
#ifdef GS_EXPERIMENT
  if (g_tune[16] == 1) return; /* timing aid (scripts/bench_lbp_stages.py): the cascade kernels alone, no rect emission */
#endif
  run_compaction(mask, cnt, nch, n, max_rects, counts,
                 LbpEmit{gc.d_scales, (unsigned)gc.scales.size(), step, rects, max_rects});
}

constexpr unsigned kLbpGroup = 8; /* frames per cascade launch (bounds mask/padded scratch) */

void launch_lbp_unpadded(const gsh_cascade *dc, const unsigned *ii, unsigned iw, unsigned ih, unsigned n,
                         unsigned *rects, unsigned *counts, unsigned max_rects, float sf, float mn,
                         float mx, int step) {
  GS_ASSERT(step > 0);
  const LbpGeomCache &gc = cascade_prepare(dc, iw, ih, sf, mn, mx, step);
  hipStream_t st = ctx().s();
  const size_t fp = (size_t)iw * ih, pp = (size_t)(iw + 1) * (ih + 1);
  for (unsigned f0 = 0; f0 < n; f0 += kLbpGroup) {
    const unsigned nn = std::min(kLbpGroup, n - f0);
    unsigned *padded = (unsigned *)ctx().scratch(SL_PAD, pp * 4 * nn);
    launch_integral_pad(dim3((iw + 64) / 64, (ih + 4) / 4, nn), st, ii + fp * f0, iw, ih, padded);
    launch_lbp_padded(dc, gc, padded, iw, ih, nn, rects + (size_t)f0 * max_rects * 4, counts + f0,
                      max_rects, step);
  }
}

/* one window on a (win_w+1) x (win_h+1) table: grid 1, block 64, lane 0 decides */
__global__ void k_lbp_single(LbpArgs a, const LbpGeom *geom, unsigned *out) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  GS_DYN_LDS(smem);
  const LbpLds t = lbp_stage_tables(smem, a, geom, threadIdx.x, 64u);
  __syncthreads();
  if (threadIdx.x == 0)
    out[0] = lbp_window_stages<true>(t, a.padded, 0u, a.limit_bytes, 0u, a.nstages) ? 1u : 0u;
}

/* ------------------------------------------------------------------ ORB host logic */

/* gs_orb_extract (ref :651-669) in the libm flavour for a list of JOBS -- a job is n frames of one size (the batch entry:
 * one job; the pyramid driver: one single-frame job per level) -- with TWO host round trips in total:
 *   (1) device: FAST + NMS + ordered emit, then the reference's selection -- stable descending sort by response (ref :639),
 *       15-px border filter, cap at nkps -- as a rank computation (k_orb_select, the kernel the GS_NO_STDLIB flavour uses),
 *       then the disc moments of the SELECTED keypoints; one copy-back of counts + records + moments;
 *       host: atan2f / sinf from libm (ref :100-101) -- the reference's angle IS whatever glibc returns, and the BRIEF
 *       bits depend on it through the (int) truncation of ref :633 -- spread over a few threads from 4096 keypoints;
 *   (2) device: BRIEF for every kept keypoint of every job, one launch per job; one copy-back of the descriptors.
 * (Round 4 copied every FAST candidate back -- up to 5000 x 48 B per frame -- and sorted on the host: 153 us per 720p
 * frame, of which ~100 were the copy and the sort.) */
/* A few parked host threads for the libm half of the ORB batch (creating a thread costs 20-50 us on these boxes, twelve of
 * them per call were a third of a 32-frame batch's time).  Workers are created on first use and parked on a condition
 * variable; run(n, fn) executes fn(0 .. n-1), the caller taking index 0; one run at a time (callers of different host
 * threads queue on the mutex).  One pool per process (host_pool() below); its destructor wakes the workers up and joins them. */
class HostPool {
 public:
  static constexpr unsigned kMax = 11;
  template <class F> void run(unsigned n, F fn) {
    if (n <= 1) {
      fn(0u);
      return;
    }
    std::lock_guard<std::mutex> one_run(run_mutex_);
    std::function<void(unsigned)> f = fn;
    {
      std::unique_lock<std::mutex> lk(m_);
      while (workers_.size() + 1 < n && workers_.size() < kMax) {
        const unsigned id = (unsigned)workers_.size() + 1u;
        workers_.emplace_back([this, id] { loop(id); });
      }
      n = std::min<unsigned>(n, (unsigned)workers_.size() + 1u);
      job_ = &f, njobs_ = n, pending_ = n - 1, gen_++;
    }
    cv_.notify_all();
    f(0u);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }
  ~HostPool() {
    {
      std::unique_lock<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
  }

 private:
  void loop(unsigned id) {
    unsigned long long seen = 0;
    for (;;) {
      std::function<void(unsigned)> *f = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        if (id < njobs_) f = job_;
      }
      if (f) {
        (*f)(id);
        std::unique_lock<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex m_, run_mutex_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  std::function<void(unsigned)> *job_ = nullptr;
  unsigned njobs_ = 0, pending_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};
/* One pool per PROCESS.  A fork()ed child (Python multiprocessing) inherits the object but none of its threads -- run() would
 * wait on `done_` forever and the destructor would join threads that do not exist -- and possibly mutexes some parent thread
 * held at that moment.  So the pool lives on the heap behind a pointer that the child's atfork handler drops (the parent's
 * object is abandoned there, not destroyed); the child builds its own on first use.  The janitor joins the workers of the
 * pool this process owns at exit / dlclose, like the function-local static did. */
static std::atomic<HostPool *> g_host_pool{nullptr};
static std::mutex *g_host_pool_make = new std::mutex; /* never destroyed: usable from any static destructor */
static void host_pool_atfork_child() {
  g_host_pool.store(nullptr, std::memory_order_relaxed); /* single-threaded here */
  g_host_pool_make = new std::mutex;
}
static struct HostPoolJanitor {
  ~HostPoolJanitor() { delete g_host_pool.exchange(nullptr); }
} g_host_pool_janitor;
HostPool &host_pool() {
  HostPool *p = g_host_pool.load(std::memory_order_acquire);
  if (!p) {
    std::lock_guard<std::mutex> lk(*g_host_pool_make);
    p = g_host_pool.load(std::memory_order_acquire);
    if (!p) {
      static const int registered = pthread_atfork(nullptr, nullptr, host_pool_atfork_child);
      (void)registered;
      p = new HostPool;
      g_host_pool.store(p, std::memory_order_release);
    }
  }
  return *p;
}

struct OrbJob {
  const uint8_t *img;
  unsigned w, h, n; /* n frames, w * h bytes apart */
  uint8_t *score;
  gs_keypoint *out; /* host: n x nkps records */
  unsigned *counts; /* host: n (may be null for single-frame jobs: see got) */
  unsigned nkps;    /* wanted per frame */
  unsigned got;     /* keypoints of frame 0 */
};

void orb_extract_libm(OrbJob *J, unsigned nj, unsigned threshold) {
  hipStream_t st = ctx().s();
  std::vector<size_t> koff(nj), foff(nj);
  size_t ktot = 0, ftot = 0, candmax = 0, cntmax = 0;
  for (unsigned j = 0; j < nj; j++) {
    J[j].got = 0;
    if (J[j].counts)
      for (unsigned f = 0; f < J[j].n; f++) J[j].counts[f] = 0;
    const bool run = J[j].nkps && J[j].n && J[j].w >= 7 && J[j].h >= 7;
    koff[j] = ktot, foff[j] = ftot;
    if (!run) {
      J[j].n = 0;
      continue;
    }
    ktot += (size_t)J[j].n * J[j].nkps, ftot += J[j].n;
    candmax = std::max(candmax, (size_t)J[j].n * std::min(J[j].nkps * 4u, 5000u));
    cntmax = std::max<size_t>(cntmax, J[j].n);
  }
  if (!ktot) return;
  unsigned *cand = (unsigned *)ctx().scratch(SL_KPS, candmax * 48 + 16);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_TOT, cntmax * 4 + 16);
  unsigned *sel = (unsigned *)ctx().scratch(SL_BEST, ktot * 48 + 16);
  unsigned *selcnt = (unsigned *)ctx().scratch(SL_LEV, ftot * 4 + 16);
  int *mom = (int *)ctx().scratch(SL_MOM, ktot * 8);
  for (unsigned j = 0; j < nj; j++) {
    const OrbJob &q = J[j];
    if (!q.n) continue;
    const unsigned cap = std::min(q.nkps * 4u, 5000u);
    const size_t fb = (size_t)q.w * q.h;
    launch_fast(q.img, q.score, q.w, q.h, q.n, cand, cnt, cap, threshold);
    GS_LAUNCH(k_orb_select, dim3(q.n), dim3(64), 0, st, (const unsigned *)cand, (const unsigned *)cnt, cap, q.w, q.h, q.nkps,
              sel + koff[j] * 12, selcnt + foff[j]);
    GS_LAUNCH(k_orient_moments, dim3(q.nkps, q.n), dim3(64), 0, st, q.img, q.w, q.h, (const unsigned *)(sel + koff[j] * 12), 12u, 15u,
              mom + koff[j] * 2, (const unsigned *)(selcnt + foff[j]), fb);
  }
  /* pinned staging (Ctx::pinned): the records come back at PCIe rate instead of through the runtime's pageable path */
  unsigned *hn = (unsigned *)ctx().pinned(Ctx::PIN_A, ftot * 4);
  unsigned *hk = (unsigned *)ctx().pinned(Ctx::PIN_B, ktot * 48);
  int *hm = (int *)ctx().pinned(Ctx::PIN_C, ktot * 8);
  GS_HIP(hipMemcpyAsync(hn, selcnt, ftot * 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(hk, sel, ktot * 48, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(hm, mom, ktot * 8, hipMemcpyDeviceToHost, st));
  ctx().sync();
  /* host half of ref :662-665: the angle and the two sines of every kept keypoint, frame by frame */
  KpIn *kin = (KpIn *)ctx().pinned(Ctx::PIN_D, ktot * sizeof(KpIn));
  struct Item { unsigned j, f; };
  std::vector<Item> items;
  size_t kept_total = 0;
  for (unsigned j = 0; j < nj; j++)
    for (unsigned f = 0; f < J[j].n; f++) items.push_back(Item{j, f}), kept_total += std::min(hn[foff[j] + f], J[j].nkps);
  auto do_items = [&](size_t a, size_t b) {
    for (size_t it = a; it < b; it++) {
      const OrbJob &q = J[items[it].j];
      const unsigned f = items[it].f, m = std::min(hn[foff[items[it].j] + f], q.nkps);
      const size_t base = koff[items[it].j] + (size_t)f * q.nkps;
      for (unsigned i = 0; i < m; i++) {
        const unsigned *r = &hk[(base + i) * 12];
        gs_keypoint &k = q.out[(size_t)f * q.nkps + i];
        k.pt.x = r[0], k.pt.y = r[1], k.response = r[2];
        k.angle = atan2f((float)hm[(base + i) * 2], (float)hm[(base + i) * 2 + 1]); /* ref :620, :100 */
        const float angle = k.angle;
        kin[base + i] = KpIn{r[0], r[1], sinf(angle), sinf((float)(angle + 1.57079f))}; /* ref :626 */
      }
    }
  };
  unsigned nthreads = 1;
  if (kept_total >= 4096 && items.size() > 1)
    nthreads = (unsigned)std::min<size_t>({HostPool::kMax + 1, std::max(1u, std::thread::hardware_concurrency()), items.size(), kept_total / 1024});
  if (nthreads > 1) host_pool().run(nthreads, [&](unsigned t) { do_items(items.size() * t / nthreads, items.size() * (t + 1) / nthreads); });
  else do_items(0, items.size());
  for (unsigned j = 0; j < nj; j++)
    for (unsigned f = 0; f < J[j].n; f++) {
      const unsigned m = std::min(hn[foff[j] + f], J[j].nkps);
      if (J[j].counts) J[j].counts[f] = m;
      if (f == 0) J[j].got = m;
    }
  if (!kept_total) return;
  KpIn *dk = (KpIn *)ctx().scratch(SL_KIN, ktot * sizeof(KpIn));
  uint32_t *dd = (uint32_t *)ctx().scratch(SL_DESC, ktot * 32);
  GS_HIP(hipMemcpyAsync(dk, kin, ktot * sizeof(KpIn), hipMemcpyHostToDevice, st));
  for (unsigned j = 0; j < nj; j++)
    if (J[j].n)
      GS_LAUNCH(k_brief, dim3(J[j].nkps, J[j].n), dim3(256), 0, st, J[j].img, J[j].w, J[j].h, (const KpIn *)(dk + koff[j]),
                dd + koff[j] * 8, (const unsigned *)(selcnt + foff[j]), (size_t)J[j].w * J[j].h);
  uint32_t *hd = (uint32_t *)hk; /* the records are consumed: their staging buffer (48 B per keypoint) takes the 32-byte descriptors */
  GS_HIP(hipMemcpyAsync(hd, dd, ktot * 32, hipMemcpyDeviceToHost, st));
  ctx().sync();
  for (unsigned j = 0; j < nj; j++)
    for (unsigned f = 0; f < J[j].n; f++) {
      const unsigned m = std::min(hn[foff[j] + f], J[j].nkps);
      const size_t base = koff[j] + (size_t)f * J[j].nkps;
      for (unsigned i = 0; i < m; i++) memcpy(J[j].out[(size_t)f * J[j].nkps + i].descriptor, &hd[(base + i) * 8], 32);
    }
}

/* the pyramid driver's view: single frames of different sizes */
struct OrbLevel {
  const uint8_t *img;
  unsigned w, h;
  uint8_t *score;
  gs_keypoint *out; /* host */
  unsigned nkps;    /* wanted */
  unsigned got;
};
void orb_extract_levels(OrbLevel *L, unsigned nl, unsigned threshold) {
  OrbJob J[4];
  for (unsigned l = 0; l < nl; l++) J[l] = OrbJob{L[l].img, L[l].w, L[l].h, 1u, L[l].score, L[l].out, nullptr, L[l].nkps, 0u};
  orb_extract_libm(J, nl, threshold);
  for (unsigned l = 0; l < nl; l++) L[l].got = J[l].got;
}

void launch_match(const uint32_t *k1, unsigned n1, const uint32_t *k2, unsigned n2,
                  unsigned *matches, unsigned *count, unsigned max_matches, float max_distance) {
  hipStream_t st = ctx().s();
  if (n1 == 0 || max_matches == 0) {
    GS_HIP(hipMemsetAsync(count, 0, 4, st));
    return;
  }
  const unsigned blocks = (n1 + 3) / 4, words = (n1 + 63) / 64;
  const unsigned nchunks = (words + kChunkWords - 1) / kChunkWords;
  unsigned long long *mask =
      (unsigned long long *)ctx().scratch(SL_MASK, (size_t)nchunks * kChunkWords * 8);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)nchunks * 4);
  unsigned *best = (unsigned *)ctx().scratch(SL_BEST, (size_t)n1 * 8);
  GS_HIP(hipMemsetAsync(mask, 0, (size_t)nchunks * kChunkWords * 8, st));
  GS_HIP(hipMemsetAsync(cnt, 0, (size_t)nchunks * 4, st));
  GS_LAUNCH(k_match, dim3(blocks), dim3(256), 0, st, k1, n1, k2, n2, max_distance, best, best + n1,
            mask, cnt);
  run_compaction(mask, cnt, nchunks, 1, max_matches, count,
                 MatchEmit{best, best + n1, matches});
}

}  // namespace

extern "C" {

void gsh_fast_score_batch(uint8_t *score, const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned threshold) {
  GS_ASSERT(score && img && w >= 7 && h >= 7 && n >= 1 && n <= kMaxZ);
  launch_fast_score(ctx().s(), img, score, w, h, n, threshold);
}
gsh_cascade *gsh_cascade_create(const struct gs_lbp_cascade *c) {
  GS_ASSERT(c && c->features && c->weak_feature_idx && c->subsets);
  ctx().ensure_device();
  gsh_cascade *dc = new gsh_cascade();
  static std::atomic<unsigned long long> next_id{1};
  dc->id = next_id.fetch_add(1);
  dc->window_w = c->window_w, dc->window_h = c->window_h;
  dc->nfeatures = c->nfeatures, dc->nweaks = c->nweaks, dc->nstages = c->nstages;
  dc->features.assign(c->features, c->features + (size_t)c->nfeatures * 4);
  dc->weak_feature_idx.assign(c->weak_feature_idx, c->weak_feature_idx + c->nweaks);
  /* device tables in EVALUATION order: stage by stage, so a stage range is one contiguous run of
   * weak classifiers (the reference indexes through stage_weak_start, ref :795-798) */
  std::vector<LbpWeak> wk;
  std::vector<LbpStage> stg(c->nstages);
  dc->weak_feature_idx.clear();
  unsigned nsub = 0;
  for (unsigned si = 0; si < c->nstages; si++) {
    stg[si] = LbpStage{(unsigned)wk.size(), c->stage_nweaks[si], c->stage_threshold[si], 0u};
    for (unsigned k = 0; k < c->stage_nweaks[si]; k++) {
      const unsigned i = (unsigned)c->stage_weak_start[si] + k;
      wk.push_back(LbpWeak{c->weak_left_val[i], c->weak_right_val[i], c->weak_subset_offset[i],
                           c->weak_num_subsets[i]});
      dc->weak_feature_idx.push_back(c->weak_feature_idx[i]);
      nsub = std::max(nsub, (unsigned)c->weak_subset_offset[i] + c->weak_num_subsets[i]);
    }
  }
  dc->nweaks = (unsigned)wk.size(); /* classifiers actually reachable through the stages */
  dc->nsub = nsub;
  /* stage truth tables (LbpStage::truth): the stage's verdict for every pattern of subset-lookup results, formed exactly
   * as ref :796-810 forms it -- float sum = 0; sum += hit ? left : right in weak order; pass = !(sum < threshold) -- in
   * IEEE float32 adds (this unit is built -ffp-contract=off; there is nothing to contract anyway) */
  std::vector<uint32_t> truth;
  for (unsigned si = 0; si < c->nstages; si++) {
    const unsigned cnt = stg[si].count;
    if (cnt < 1 || cnt > 12) continue;
    const unsigned words = std::max(1u, (1u << cnt) / 32u), off = (unsigned)truth.size();
    truth.resize(off + words, 0u);
    for (unsigned pat = 0; pat < (1u << cnt); pat++) {
      volatile float sum = 0.0f; /* volatile: one rounded float32 add per classifier, whatever the optimiser thinks */
      for (unsigned k = 0; k < cnt; k++) {
        const LbpWeak &q = wk[stg[si].first + k];
        sum = sum + (((pat >> k) & 1u) ? q.left : q.right);
      }
      if (!(sum < stg[si].threshold)) truth[off + (pat >> 5)] |= 1u << (pat & 31u);
    }
    stg[si].truth = off + 1u;
  }
  dc->ntruth = (unsigned)truth.size();
  GS_HIP(hipMalloc((void **)&dc->d_truth, std::max<size_t>(1, truth.size()) * 4));
  if (!truth.empty()) GS_HIP(hipMemcpy(dc->d_truth, truth.data(), truth.size() * 4, hipMemcpyHostToDevice));
  GS_HIP(hipMalloc((void **)&dc->d_weak, std::max<size_t>(1, wk.size()) * sizeof(LbpWeak)));
  GS_HIP(hipMalloc((void **)&dc->d_stage, std::max<size_t>(1, stg.size()) * sizeof(LbpStage)));
  GS_HIP(hipMalloc((void **)&dc->d_subsets, std::max<size_t>(1, nsub) * 4));
  GS_HIP(hipMemcpy(dc->d_weak, wk.data(), wk.size() * sizeof(LbpWeak), hipMemcpyHostToDevice));
  GS_HIP(hipMemcpy(dc->d_stage, stg.data(), stg.size() * sizeof(LbpStage), hipMemcpyHostToDevice));
  GS_HIP(hipMemcpy(dc->d_subsets, c->subsets, (size_t)nsub * 4, hipMemcpyHostToDevice));
  return dc;
}
void gsh_cascade_destroy(gsh_cascade *dc) {
  if (!dc) return;
  ctx().sync();
  /* geometry tables of this handle in the calling thread's cache go with it; other threads' entries
   * are keyed by the (never reused) id and are dropped when their context is released */
  auto it = ctx().geom_cache.find(dc->id);
  if (it != ctx().geom_cache.end()) {
    if (it->second.d_scales) (void)hipFree(it->second.d_scales);
    if (it->second.d_geom) (void)hipFree(it->second.d_geom);
    ctx().geom_cache.erase(it);
  }
  gsh_cascade_tables_deleter()(dc);
}
void gsh_lbp_detect_batch(const gsh_cascade *dc, const unsigned *ii, unsigned iw, unsigned ih,
                          unsigned n, struct gs_rect *rects, unsigned *counts, unsigned max_rects,
                          float scale_factor, float min_scale, float max_scale, int step) {
  GS_ASSERT(dc && ii && rects && counts && iw > 0 && ih > 0);
  launch_lbp_unpadded(dc, ii, iw, ih, n, (unsigned *)rects, counts,
                      max_rects, scale_factor, min_scale, max_scale, step);
}
void gsh_lbp_count_evaluated(unsigned long long *counter_dev) { g_lbp_evaluated = counter_dev; }
uint64_t gsh_lbp_window_count(const struct gs_lbp_cascade *c, unsigned iw, unsigned ih,
                              float scale_factor, float min_scale, float max_scale, int step) {
  gsh_cascade tmp;
  tmp.window_w = c->window_w, tmp.window_h = c->window_h, tmp.nweaks = 0;
  std::vector<LbpScale> sc;
  std::vector<LbpGeom> ge;
  bool guard;
  unsigned long long nwin;
  build_scales(tmp, iw, ih, scale_factor, min_scale, max_scale, step, sc, ge, guard, nwin);
  return nwin;
}

/* ---------------------------------------------------------------- batch: FAST / ORB / match */
void gsh_fast_batch(const uint8_t *img, uint8_t *scoremap, unsigned w, unsigned h, unsigned n,
                    struct gs_keypoint *kps, unsigned *counts, unsigned nkps, unsigned threshold) {
  GS_ASSERT(img && scoremap && kps && counts && nkps > 0 && w > 0 && h > 0);
  launch_fast(img, scoremap, w, h, n, (unsigned *)kps, counts, nkps, threshold);
}
unsigned gsh_orb_extract(const uint8_t *img_dev, unsigned w, unsigned h, uint8_t *scoremap_dev,
                         struct gs_keypoint *kps_host, unsigned nkps, unsigned threshold) {
  GS_ASSERT(img_dev && scoremap_dev && kps_host && nkps > 0 && w > 0 && h > 0);
  OrbLevel L{img_dev, w, h, scoremap_dev, kps_host, nkps, 0};
  orb_extract_levels(&L, 1, threshold);
  return L.got;
}
/* gs_orb_extract (ref :651-669), libm flavour, for n frames of one size: orb_extract_libm with one job */
void gsh_orb_extract_batch(const uint8_t *img_dev, unsigned w, unsigned h, unsigned n,
                           uint8_t *scoremap_dev, struct gs_keypoint *kps_host, unsigned *counts_host,
                           unsigned nkps, unsigned threshold) {
  GS_ASSERT(img_dev && scoremap_dev && kps_host && counts_host && nkps > 0 && w > 0 && h > 0);
  constexpr unsigned kOrbGroup = 4096; /* frames per pass: bounds the host-side buffers and grid.y */
  for (unsigned f0 = 0; f0 < n; f0 += kOrbGroup) {
    OrbJob q{img_dev + (size_t)w * h * f0, w, h, std::min(kOrbGroup, n - f0), scoremap_dev + (size_t)w * h * f0,
             kps_host + (size_t)f0 * nkps, counts_host + f0, nkps, 0u};
    orb_extract_libm(&q, 1, threshold);
  }
}

/* gs_orb_extract (ref :651-669) for n frames, everything on the device, no host round trip: FAST ->
 * selection (stable descending sort + 15-px border filter + cap, k_orb_select) -> orientation + BRIEF
 * (k_orb_describe) with the reference's GS_NO_STDLIB trig (ref :70-88).  Results are those of the
 * reference header compiled with -DGS_NO_STDLIB; the libm flavour (glibc atan2f / sinf bits) stays
 * with gsh_orb_extract_batch, whose trig runs on the host. */
void gsh_orb_extract_batch_nostdlib(const uint8_t *img_dev, unsigned w, unsigned h, unsigned n,
                                    uint8_t *scoremap_dev, struct gs_keypoint *kps_dev, unsigned *counts_dev,
                                    unsigned nkps, unsigned threshold) {
  GS_ASSERT(img_dev && scoremap_dev && kps_dev && counts_dev && nkps > 0 && w > 0 && h > 0);
  if (n == 0) return;
  hipStream_t st = ctx().s();
  if (w < 7 || h < 7) {
    GS_HIP(hipMemsetAsync(counts_dev, 0, (size_t)n * 4, st));
    return;
  }
  const size_t fb = (size_t)w * h;
  const unsigned cap = std::min(nkps * 4u, 5000u);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    unsigned *cand = (unsigned *)ctx().scratch(SL_KPS, (size_t)nn * cap * 48 + 16);
    unsigned *cnt = (unsigned *)ctx().scratch(SL_TOT, (size_t)nn * 4 + 16);
    launch_fast(img_dev + fb * f0, scoremap_dev + fb * f0, w, h, nn, cand, cnt, cap, threshold);
    unsigned *out = (unsigned *)kps_dev + (size_t)f0 * nkps * 12;
    GS_LAUNCH(k_orb_select, dim3(nn), dim3(64), 0, st, (const unsigned *)cand, (const unsigned *)cnt, cap, w, h, nkps,
              out, counts_dev + f0);
    GS_LAUNCH(k_orb_describe, dim3(nkps, nn), dim3(256), 0, st, img_dev + fb * f0, w, h, fb, out,
              (const unsigned *)(counts_dev + f0), nkps);
  }
}

size_t gsh_orb_pyramid_buffer_bytes(unsigned w, unsigned h, unsigned n_levels) {
  if (n_levels > 4) n_levels = 4;
  size_t levels = 0, maps = (size_t)w * h;
  for (unsigned l = 1; l < n_levels; l++) {
    w /= 2, h /= 2;
    if (w < 32 || h < 32) break;
    levels += (size_t)w * h, maps += (size_t)w * h;
  }
  return levels + maps;
}
/* ref examples/nanomagick/nanomagick.c:245-290 with every level resident on the device */
unsigned gsh_orb_extract_pyramid(const uint8_t *img_dev, unsigned w, unsigned h, uint8_t *buffer_dev,
                                 struct gs_keypoint *kps_host, unsigned nkps, unsigned threshold,
                                 unsigned n_levels) {
  GS_ASSERT(img_dev && buffer_dev && kps_host && w > 0 && h > 0);
  if (n_levels > 4) n_levels = 4;
  if (n_levels == 0 || nkps == 0) return 0; /* the reference driver's level loop is empty (nanomagick.c:262) */
  const uint8_t *lev[4];
  unsigned lw[4], lh[4], total = 0;
  size_t off = 0;
  lev[0] = img_dev, lw[0] = w, lh[0] = h;
  for (unsigned l = 1; l < n_levels; l++) {
    const unsigned nw = lw[l - 1] / 2, nh = lh[l - 1] / 2;
    if (nw < 32 || nh < 32) {
      n_levels = l;
      break;
    }
    uint8_t *d = buffer_dev + off;
    off += (size_t)nw * nh;
    gsh_downsample_batch(d, lev[l - 1], lw[l - 1], lh[l - 1], 1);
    lev[l] = d, lw[l] = nw, lh[l] = nh;
  }
  /* per-level quotas depend on how many keypoints the earlier levels produced only for the last
   * level ("the remainder", nanomagick.c:275): run the first n_levels-1 levels as one batch, then
   * the last one */
  OrbLevel L[4];
  uint8_t *sm[4];
  for (unsigned l = 0; l < n_levels; l++) {
    sm[l] = buffer_dev + off;
    off += (size_t)lw[l] * lh[l];
  }
  const unsigned per = nkps / n_levels;
  for (unsigned l = 0; l + 1 < n_levels; l++) L[l] = OrbLevel{lev[l], lw[l], lh[l], sm[l], kps_host + (size_t)l * per, per, 0};
  if (n_levels > 1) orb_extract_levels(L, n_levels - 1, threshold);
  for (unsigned l = 0; l + 1 < n_levels; l++) {
    /* levels write at l*per; the reference packs them back to back (total_kps) */
    if (L[l].got && total != l * per) memmove(kps_host + total, kps_host + (size_t)l * per, (size_t)L[l].got * sizeof(gs_keypoint));
    for (unsigned i = total; i < total + L[l].got; i++) kps_host[i].pt.x <<= l, kps_host[i].pt.y <<= l;
    total += L[l].got;
  }
  {
    const unsigned l = n_levels - 1, want = nkps - total;
    if (want) {
      L[l] = OrbLevel{lev[l], lw[l], lh[l], sm[l], kps_host + total, want, 0};
      orb_extract_levels(&L[l], 1, threshold);
      for (unsigned i = total; i < total + L[l].got; i++) kps_host[i].pt.x <<= l, kps_host[i].pt.y <<= l;
      total += L[l].got;
    }
  }
  return total;
}
void gsh_match_orb_dev(const struct gs_keypoint *k1, unsigned n1, const struct gs_keypoint *k2,
                       unsigned n2, struct gs_match *matches, unsigned *count,
                       unsigned max_matches, float max_distance) {
  GS_ASSERT(k1 && k2 && matches && count);
  launch_match((const uint32_t *)k1, n1, (const uint32_t *)k2, n2, (unsigned *)matches, count,
               max_matches, max_distance);
}

/* The reference re-reads the caller's tables on every call (ref :790-835), so a cascade edited in
 * place, or freed and rebuilt at the same addresses, must take effect.  The flattened device copy
 * is therefore cached per calling thread keyed by a hash of the table CONTENTS (FNV-1a 64 over
 * ~7 KB: microseconds next to the launch), never by the struct's pointer values. */
static uint64_t cascade_content_hash(const struct gs_lbp_cascade *c) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  const uint16_t dims[5] = {c->window_w, c->window_h, c->nfeatures, c->nweaks, c->nstages};
  mix(dims, sizeof dims);
  mix(c->features, (size_t)c->nfeatures * 4);
  mix(c->weak_feature_idx, (size_t)c->nweaks * 2);
  mix(c->weak_left_val, (size_t)c->nweaks * 4);
  mix(c->weak_right_val, (size_t)c->nweaks * 4);
  mix(c->weak_subset_offset, (size_t)c->nweaks * 2);
  mix(c->weak_num_subsets, (size_t)c->nweaks * 2);
  unsigned nsub = 0;
  for (unsigned i = 0; i < c->nweaks; i++)
    nsub = std::max(nsub, (unsigned)c->weak_subset_offset[i] + c->weak_num_subsets[i]);
  mix(c->subsets, (size_t)nsub * 4);
  mix(c->stage_weak_start, (size_t)c->nstages * 2);
  mix(c->stage_nweaks, (size_t)c->nstages * 2);
  mix(c->stage_threshold, (size_t)c->nstages * 4);
  return h;
}
static gsh_cascade *cached_cascade(const struct gs_lbp_cascade *c) {
  Ctx &cx = ctx();
  const uint64_t h = cascade_content_hash(c);
  if (cx.dropin_cascade && cx.dropin_cascade_hash == h) return cx.dropin_cascade;
  if (cx.dropin_cascade) gsh_cascade_destroy(cx.dropin_cascade);
  cx.dropin_cascade = gsh_cascade_create(c);
  cx.dropin_cascade_hash = h;
  return cx.dropin_cascade;
}

unsigned gs_lbp_detect(const struct gs_lbp_cascade *c, const unsigned *ii, unsigned iw,
                       unsigned ih, struct gs_rect *rects, unsigned max_rects, float scale_factor,
                       float min_scale, float max_scale, int step) { /* ref :815 */
  GS_ASSERT(c && ii && iw > 0 && ih > 0 && (rects || max_rects == 0));
  if (max_rects == 0) return 0;
  gsh_cascade *dc = cached_cascade(c);
  const size_t np = (size_t)iw * ih;
  const unsigned *dii = (const unsigned *)stage_in(ii, np * 4, SL_II);
  const bool rhost = !is_dev(rects);
  unsigned *dr = rhost ? (unsigned *)ctx().scratch(SL_OUT, (size_t)max_rects * 16) : (unsigned *)rects;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  launch_lbp_unpadded(dc, dii, iw, ih, 1, dr, dcnt, max_rects, scale_factor, min_scale, max_scale,
                      step);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  if (rhost && n) GS_HIP(hipMemcpy(rects, dr, (size_t)n * 16, hipMemcpyDeviceToHost));
  return n;
}

unsigned gs_lbp_window(const struct gs_lbp_cascade *c, const unsigned *ii, unsigned iw, unsigned ih,
                       int x, int y, float scale) { /* ref :790 */
  GS_ASSERT(c && ii);
  const int win_w = (int)((int)c->window_w * scale), win_h = (int)((int)c->window_h * scale);
  if (x + win_w > (int)iw || y + win_h > (int)ih) return 0; /* ref :793 */
  if (x < 0 || y < 0 || win_w <= 0 || win_h <= 0) return 0;
  gsh_cascade *dc = cached_cascade(c);
  hipStream_t st = ctx().s();
  /* (win_w+1) x (win_h+1) zero-bordered sub-table around the window: the cascade only ever
   * forms D + A - B - C differences, so absolute table values carry over unchanged */
  const unsigned S = (unsigned)win_w + 1, R = (unsigned)win_h + 1;
  unsigned *tab = (unsigned *)ctx().scratch(SL_PAD, (size_t)S * R * 4 + 64);
  GS_HIP(hipMemsetAsync(tab, 0, (size_t)S * R * 4, st));
  const unsigned cx = x > 0 ? 1 : 0, cy = y > 0 ? 1 : 0;
  const unsigned *src0 = ii + (size_t)(y - (int)cy) * iw + (x - (int)cx);
  GS_HIP(hipMemcpy2DAsync(tab + (size_t)(1 - cy) * S + (1 - cx), (size_t)S * 4, src0, (size_t)iw * 4,
                          (size_t)(win_w + cx) * 4, (size_t)(win_h + cy),
                          is_dev(ii) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  std::vector<LbpGeom> geom(dc->nweaks);
  for (unsigned wi = 0; wi < dc->nweaks; wi++) {
    const int fi = dc->weak_feature_idx[wi];
    int fx = (int)((int)dc->features[fi * 4 + 0] * scale), fy = (int)((int)dc->features[fi * 4 + 1] * scale);
    int fw = (int)((int)dc->features[fi * 4 + 2] * scale), fh = (int)((int)dc->features[fi * 4 + 3] * scale);
    if (fw < 1) fw = 1;
    if (fh < 1) fh = 1;
    geom[wi] = LbpGeom{(fy * (int)S + fx) * 4, fw * 4, fh * (int)S * 4, 0};
  }
  LbpGeom *dg = (LbpGeom *)ctx().scratch(SL_TAB, std::max<size_t>(1, geom.size()) * sizeof(LbpGeom));
  GS_HIP(hipMemcpyAsync(dg, geom.data(), geom.size() * sizeof(LbpGeom), hipMemcpyHostToDevice, st));
  unsigned *dout = (unsigned *)ctx().scratch(SL_TOT, 16);
  LbpArgs a;
  memset(&a, 0, sizeof a);
  a.padded = tab, a.frame_stride = (size_t)S * R, a.S = S, a.limit_bytes = (unsigned)(((size_t)S * R - 1) * 4), a.step = 1;
  a.nweaks = dc->nweaks, a.nstages = dc->nstages, a.nsub = dc->nsub;
  a.weak = dc->d_weak, a.stage = dc->d_stage, a.subsets = dc->d_subsets;
  const size_t lds = (size_t)dc->nstages * sizeof(LbpStage) +
                     (size_t)dc->nweaks * (sizeof(LbpWeak) + sizeof(LbpGeom)) + (size_t)dc->nsub * 4;
  GS_LAUNCH(k_lbp_single, dim3(1), dim3(64), lds, st, a, (const LbpGeom *)dg, dout);
  unsigned r = 0;
  GS_HIP(hipMemcpyAsync(&r, dout, 4, hipMemcpyDeviceToHost, st));
  ctx().sync();
  return r;
}

unsigned gs_fast(struct gs_image img, struct gs_image scoremap, struct gs_keypoint *kps,
                 unsigned nkps, unsigned threshold) { /* ref :482 */
  GS_ASSERT(GS_VALID(img) && kps && nkps > 0);
  const unsigned w = img.w, h = img.h;
  if (w < 7 || h < 7) return 0;
  /* The reference writes the map through gs_set and reads it through gs_get (ref :512, :518-524), so a
   * map of another size -- or no map at all -- is legal: positions outside it are never written and
   * read 0.  Same here: the kernels run on an image-sized device map M that starts as the caller's
   * map where the two overlap (0 elsewhere); positions outside the caller's map are zeroed again
   * between the two passes; the overlap is copied back. */
  if (!GS_VALID(scoremap)) return 0; /* every gs_get(scoremap) is 0: the NMS pass skips every pixel */
  const size_t nb = (size_t)w * h;
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool same = scoremap.w == w && scoremap.h == h;
  const bool mhost = !is_dev(scoremap.data);
  const unsigned ow = std::min(w, scoremap.w), oh = std::min(h, scoremap.h); /* overlap */
  uint8_t *dm = (mhost || !same) ? (uint8_t *)ctx().scratch(SL_AUX, nb) : scoremap.data;
  /* NMS reads the caller's 3-px frame (ref :524): ship the whole map in */
  if (same) {
    if (mhost) GS_HIP(hipMemcpyAsync(dm, scoremap.data, nb, hipMemcpyHostToDevice, ctx().s()));
  } else {
    GS_HIP(hipMemsetAsync(dm, 0, nb, ctx().s()));
    GS_HIP(hipMemcpy2DAsync(dm, w, scoremap.data, scoremap.w, ow, oh,
                            mhost ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx().s()));
  }
  const bool khost = !is_dev(kps);
  unsigned *dk = khost ? (unsigned *)ctx().scratch(SL_KPS, (size_t)nkps * 48) : (unsigned *)kps;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  launch_fast(s, dm, w, h, 1, dk, dcnt, nkps, threshold, same ? 0u : scoremap.w, same ? 0u : scoremap.h);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, ctx().s()));
  if (same) {
    if (mhost) GS_HIP(hipMemcpyAsync(scoremap.data, dm, nb, hipMemcpyDeviceToHost, ctx().s()));
  } else {
    GS_HIP(hipMemcpy2DAsync(scoremap.data, scoremap.w, dm, w, ow, oh,
                            mhost ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx().s()));
  }
  ctx().sync();
  if (khost && n) GS_HIP(hipMemcpy(kps, dk, (size_t)n * 48, hipMemcpyDeviceToHost));
  return n;
}

/* image patch [x-r, x+r] x [y-r, y+r] zero-filled outside the image, on the device */
static const uint8_t *stage_patch(struct gs_image img, int x, int y, int r, int slot) {
  const int side = 2 * r + 1;
  uint8_t *d = (uint8_t *)ctx().scratch(slot, (size_t)side * side);
  hipStream_t st = ctx().s();
  GS_HIP(hipMemsetAsync(d, 0, (size_t)side * side, st));
  const int xa = std::max(0, x - r), xb = std::min((int)img.w - 1, x + r);
  const int ya = std::max(0, y - r), yb = std::min((int)img.h - 1, y + r);
  if (xa <= xb && ya <= yb)
    GS_HIP(hipMemcpy2DAsync(d + (size_t)(ya - (y - r)) * side + (xa - (x - r)), side,
                            img.data + (size_t)ya * img.w + xa, img.w, (size_t)(xb - xa + 1),
                            (size_t)(yb - ya + 1),
                            is_dev(img.data) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  return d;
}

/* the two moments of ref :610-619 as floats (exact integers for r <= kOrientExactR, the reference's own float32
 * accumulation order beyond) */
static void orientation_moments(struct gs_image img, unsigned x, unsigned y, unsigned r, float &m01, float &m10);

float gs_compute_orientation(struct gs_image img, unsigned x, unsigned y, unsigned r) { /* ref :608 */
  GS_ASSERT(GS_VALID(img) && x >= r && y >= r && x < img.w - r && y < img.h - r);
  float m01, m10;
  orientation_moments(img, x, y, r, m01, m10);
  return atan2f(m01, m10); /* ref :620 -> libm, as the reference (ref :100) */
}
/* The reference's other trig flavour: built -DGS_NO_STDLIB its gs_atan2 / gs_sin are the float32 polynomials of
 * ref :70-88 (what examples/wasm/grayskull.c:31-35 ships) and gs_assert is compiled out (ref :69).  include/grayskull.h
 * binds gs_compute_orientation / gs_brief_descriptor / gs_orb_extract to these three symbols under the same macro, so a
 * caller built that way keeps bit-for-bit parity with ITS reference build.  No precondition aborts, like ref :69. */
float gs_compute_orientation_nostdlib(struct gs_image img, unsigned x, unsigned y, unsigned r) {
  /* no precondition on x, y, r: the reference built this way has no gs_assert (ref :69) and still forms the moments, its
   * gs_get reading 0 outside the image (ref :41-43; x + dx wraps as unsigned and fails the range test) -- stage_patch
   * zero-fills the same pixels.  Only an invalid image (nothing to read at all) returns early. */
  if (!GS_VALID(img)) return 0.0f;
  /* x, y act as signed there (x + dx wraps): a centre further than r outside the image sees no pixel at all */
  const long long xs = (int)x, ys = (int)y, rr = r;
  if (xs + rr < 0 || ys + rr < 0 || xs - rr >= (long long)img.w || ys - rr >= (long long)img.h) return gs_atan2_poly(0.0f, 0.0f);
  float m01, m10;
  orientation_moments(img, x, y, r, m01, m10);
  return gs_atan2_poly(m01, m10);
}
static void orientation_moments(struct gs_image img, unsigned x, unsigned y, unsigned r, float &m01, float &m10) {
  hipStream_t st = ctx().s();
  const uint8_t *patch = stage_patch(img, (int)x, (int)y, (int)r, SL_AUX2);
  if (r > kOrientExactR) { /* partial sums can pass 2^24: the reference's float32 order, one thread */
    float *df = (float *)ctx().scratch(SL_MOM, 16);
    GS_LAUNCH(k_orient_moments_seq, dim3(1), dim3(1), 0, st, patch, 2 * r + 1, 2 * r + 1, r, r, r, df);
    float mf[2];
    GS_HIP(hipMemcpyAsync(mf, df, 8, hipMemcpyDeviceToHost, st));
    ctx().sync();
    m01 = mf[0], m10 = mf[1];
    return;
  }
  unsigned pt[2] = {r, r};
  unsigned *dp = (unsigned *)ctx().scratch(SL_KIN, 16);
  int *dm = (int *)ctx().scratch(SL_MOM, 16);
  GS_HIP(hipMemcpyAsync(dp, pt, 8, hipMemcpyHostToDevice, st));
  GS_LAUNCH(k_orient_moments, dim3(1), dim3(64), 0, st, patch, 2 * r + 1, 2 * r + 1,
            (const unsigned *)dp, 2u, r, dm, (const unsigned *)nullptr);
  int m[2];
  GS_HIP(hipMemcpyAsync(m, dm, 8, hipMemcpyDeviceToHost, st));
  ctx().sync();
  m01 = (float)m[0], m10 = (float)m[1];
}

static void brief_descriptor(struct gs_image img, struct gs_keypoint *kp, bool poly);
void gs_brief_descriptor(struct gs_image img, struct gs_keypoint *kp) { /* ref :623 */
  GS_ASSERT(GS_VALID(img) && kp);
  brief_descriptor(img, kp, false);
}
void gs_brief_descriptor_nostdlib(struct gs_image img, struct gs_keypoint *kp) { /* ref :623 built -DGS_NO_STDLIB */
  if (!(GS_VALID(img) && kp)) return;
  brief_descriptor(img, kp, true);
}
static void brief_descriptor(struct gs_image img, struct gs_keypoint *kp, bool poly) {
  hipStream_t st = ctx().s();
  const int R = 22; /* |pattern| <= 15 rotated reaches <= 21 px */
  gs_keypoint k;
  if (is_dev(kp)) gsh_download(&k, kp, sizeof k);
  else k = *kp;
  const uint8_t *patch = stage_patch(img, (int)k.pt.x, (int)k.pt.y, R, SL_AUX2);
  const float angle = k.angle;
  KpIn in{(unsigned)R, (unsigned)R, poly ? gs_sin_poly(angle) : sinf(angle),
          poly ? gs_sin_poly((float)(angle + 1.57079f)) : sinf((float)(angle + 1.57079f))}; /* ref :626 */
  KpIn *dk = (KpIn *)ctx().scratch(SL_KIN, sizeof in);
  uint32_t *dd = (uint32_t *)ctx().scratch(SL_DESC, 32);
  GS_HIP(hipMemcpyAsync(dk, &in, sizeof in, hipMemcpyHostToDevice, st));
  GS_LAUNCH(k_brief, dim3(1), dim3(256), 0, st, patch, 2u * R + 1, 2u * R + 1, (const KpIn *)dk, dd);
  GS_HIP(hipMemcpyAsync(k.descriptor, dd, 32, hipMemcpyDeviceToHost, st));
  ctx().sync();
  if (is_dev(kp)) gsh_upload(kp, &k, sizeof k);
  else memcpy(kp->descriptor, k.descriptor, 32);
}

/* gs_orb_extract of a reference built -DGS_NO_STDLIB (ref :651-669 with the trig of :70-88): the device-resident path
 * (FAST -> k_orb_select -> k_orb_describe, no host round trip between them); host buffers are staged like everywhere. */
unsigned gs_orb_extract_nostdlib(struct gs_image img, struct gs_keypoint *kps, unsigned nkps,
                                 unsigned threshold, uint8_t *scoremap_buffer) {
  if (!(GS_VALID(img) && kps && nkps > 0 && scoremap_buffer)) return 0; /* gs_assert is compiled out (ref :69) */
  const unsigned w = img.w, h = img.h;
  if (w < 7 || h < 7) return 0;
  const size_t nb = (size_t)w * h;
  hipStream_t st = ctx().s();
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool mhost = !is_dev(scoremap_buffer), khost = !is_dev(kps);
  uint8_t *dm = mhost ? (uint8_t *)ctx().scratch(SL_AUX, nb) : scoremap_buffer;
  if (mhost) GS_HIP(hipMemcpyAsync(dm, scoremap_buffer, nb, hipMemcpyHostToDevice, st));
  gs_keypoint *dk = khost ? (gs_keypoint *)ctx().scratch(SL_DESC, (size_t)nkps * sizeof(gs_keypoint)) : kps;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_KIN, 16); /* slots the batch entry below does not touch */
  gsh_orb_extract_batch_nostdlib(s, w, h, 1, dm, dk, dcnt, nkps, threshold);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, st));
  if (mhost) GS_HIP(hipMemcpyAsync(scoremap_buffer, dm, nb, hipMemcpyDeviceToHost, st));
  ctx().sync();
  if (khost && n) GS_HIP(hipMemcpy(kps, dk, (size_t)n * sizeof(gs_keypoint), hipMemcpyDeviceToHost));
  return n;
}

unsigned gs_orb_extract(struct gs_image img, struct gs_keypoint *kps, unsigned nkps,
                        unsigned threshold, uint8_t *scoremap_buffer) { /* ref :651 */
  GS_ASSERT(GS_VALID(img) && kps && nkps > 0 && scoremap_buffer);
  const unsigned w = img.w, h = img.h;
  if (w < 7 || h < 7) return 0;
  const size_t nb = (size_t)w * h;
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool mhost = !is_dev(scoremap_buffer);
  uint8_t *dm = mhost ? (uint8_t *)ctx().scratch(SL_AUX, nb) : scoremap_buffer;
  if (mhost) GS_HIP(hipMemcpyAsync(dm, scoremap_buffer, nb, hipMemcpyHostToDevice, ctx().s()));
  const bool khost = !is_dev(kps);
  std::vector<gs_keypoint> tmp;
  gs_keypoint *out = kps;
  if (!khost) {
    tmp.resize(nkps);
    out = tmp.data();
  }
  OrbLevel L{s, w, h, dm, out, nkps, 0};
  orb_extract_levels(&L, 1, threshold);
  const unsigned n = L.got;
  if (mhost) GS_HIP(hipMemcpyAsync(scoremap_buffer, dm, nb, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  if (!khost && n) gsh_upload(kps, out, (size_t)n * sizeof(gs_keypoint));
  return n;
}

unsigned gs_match_orb(const struct gs_keypoint *kps1, unsigned n1, const struct gs_keypoint *kps2,
                      unsigned n2, struct gs_match *matches, unsigned max_matches,
                      float max_distance) { /* ref :680 */
  GS_ASSERT(kps1 && kps2 && matches);
  if (n1 == 0 || max_matches == 0) return 0;
  const uint32_t *d1 = (const uint32_t *)stage_in(kps1, (size_t)n1 * 48, SL_IN);
  const uint32_t *d2 = n2 ? (const uint32_t *)stage_in(kps2, (size_t)n2 * 48, SL_AUX)
                          : (const uint32_t *)ctx().scratch(SL_AUX, 48);
  const bool mhost = !is_dev(matches);
  unsigned *dm = mhost ? (unsigned *)ctx().scratch(SL_OUT, (size_t)max_matches * 12) : (unsigned *)matches;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  launch_match(d1, n1, d2, n2, dm, dcnt, max_matches, max_distance);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  if (mhost && n) GS_HIP(hipMemcpy(matches, dm, (size_t)n * 12, hipMemcpyDeviceToHost));
  return n;
}

}  /* extern "C" */
