/*
 * gs_stencil.cpp -- launchers and C ABI of the pixel-plane operations of libgrayskull_hip.so: gs_blur / gs_sobel /
 * gs_erode / gs_dilate, histogram / Otsu / threshold, the fused config-2 pipeline, gs_integral, adaptive threshold /
 * filter / downsample, geometry and template matching, the synthetic-frame generator and the checksums.  Both the
 * reference's own names and signatures (include/grayskull.h, each citing the reference definition it replaces) and the
 * device-resident batch entry points of include/grayskull_hip.h.  Every compute step is a HIP kernel from k_*.h.
 */
#include "gs_internal.h"

#include "k_geom.h"
#include "k_integral.h"
#include "k_pointwise.h"
#include "k_stencil.h"
#include "k_tmatch.h"

namespace gsi {

/* rows: output rows per frame.  One wave per (1024-px column block, band, frame).
 * HBM-bound per-call kernels (waves_per_simd >= 5) take SHORT bands of 8 rows (more for wide halos):
 * blocks are dispatched in band order, so the few thousand waves that are resident at any time work
 * on neighbouring rows of a few frames -- the DRAM pages they stream through are shared and the
 * halo rows of the next band are still in the Infinity Cache -- instead of each wave streaming its
 * own distant band.  Measured (profiles/r02g_small_bands_*.log, 64 frames): gs_sobel 4096x4096
 * 4.87 -> 5.62 TB/s (0.61 -> 0.70 of 8 TB/s) although it re-reads 2 halo rows per 8, 3840x2160
 * 4.76 -> 5.23; gs_blur(2) 4.88 -> 5.34 / 4.65 -> 4.90; erode 5.05 -> 5.57; plain strip copy 5.08 ->
 * 5.59.  Round 1 sized bands so that every wave of the launch was resident at once (20 bands of
 * 108 rows for 64 4K frames) and never tried bands below 16 rows.
 * The VALU-heavy fused kernels (waves_per_simd 3) keep long bands: each band first recomputes
 * 2R+2 rows of horizontal sums (8-row bands: 0.25 -> 0.38 ms per 64 frames). */
StripCfg strip_cfg(unsigned w, unsigned rows, unsigned n, unsigned waves_per_simd, unsigned halo_rows, unsigned short_T, int rg) {
  StripCfg c;
  /* lanes to place (k_strip.h): ragged rows may idle lane 0, the realigning flavour uses the lane behind the last one.
   * rg = -1: the caller's kernel has no flavours (or ignores the extra lane) */
  const unsigned strips = (w + 15) / 16 + (rg == 2 ? strip_realign_shift(w) + strip_realign_helper(w) : rg == 1 ? strip_ragged_shift(w) : (w & 15u) ? 1u : 0u);
  const unsigned long long waves_x = (strips + 63) / 64;
  unsigned long long t;
  if (g_tune[0] > 0) {
    t = (unsigned long long)g_tune[0];
  } else if (waves_per_simd >= 5) {
    /* round 3, with the XCD-aware band mapping (a band's halo rows were just fetched by its own XCD) even shorter bands
     * pay: gs_sobel 6 rows (4096^2 0.706 -> 0.720 of the peak, 512 x 4K 0.705 -> 0.722; odd heights break its 2-row
     * unroll groups), gs_blur(2) 6 (0.671 -> 0.703 / 0.692 -> 0.705), gs_blur(1) and the morphology 4 (0.707 -> 0.730,
     * 0.716 -> 0.738 / 0.728 -> 0.746); profiles/r03l_strip_band_height_xcd.log.  short_T is the caller's choice. */
    t = std::max(short_T, 2u * halo_rows);
  } else {
    /* bands per frame: the launch should fill the chip's resident-wave capacity a whole number of
     * times.  One round when the batch is small enough; for big batches (512 4K frames: 2048 wave
     * columns against 3072 slots at 3 waves per SIMD) one band per frame would leave a third of the
     * chip idle for the whole launch and two bands would run a 1.33rd round at a third occupancy, so
     * take the band count whose last round is fullest (here 3: exactly two rounds; gsh_blur_sobel_batch
     * 512 frames 1.96 -> 1.89 ms, 200 frames 0.90 -> 0.74 ms, profiles/r02f_band_count_512.log). */
    const unsigned long long cap = (unsigned long long)topo().simds() * waves_per_simd, wn = waves_x * n;
    const unsigned long long nb_max = rows / 8 ? rows / 8 : 1; /* bands of >= 8 rows */
    unsigned long long nb = cap / wn;
    if (nb < 1) nb = 1;
    if (nb > nb_max) nb = nb_max;
    /* (only for the VALU-heavy fused kernels, which run 3 waves per SIMD: the HBM-bound per-call
     * kernels lose 3-6 % to the extra halo rows of shorter bands, profiles/r02f_band_count_512.log) */
    if (waves_per_simd <= 3 && nb < 4 && wn * nb * 8 < cap * 7) { /* a lone round under 7/8 full: try 2..6 rounds */
      double best = (double)((wn * nb + cap - 1) / cap) * cap / (double)(wn * nb);
      for (unsigned long long c = nb + 1; c <= nb_max && c <= nb + 6; c++) {
        const double waste = (double)((wn * c + cap - 1) / cap) * cap / (double)(wn * c);
        if (waste < best - 0.02) best = waste, nb = c;
      }
    }
    t = (rows + nb - 1) / nb;
  }
  c.T = (unsigned)t;
  const unsigned nb = (rows + c.T - 1) / c.T;
  /* block shape: 256 threads = 4 waves; a wave is 1024 px of a row.  Frames narrower than 4096 px put the spare
   * waves on further BANDS (64 x 4 up to 1024 px, 128 x 2 up to 2048 px) instead of columns that do not exist --
   * a 1920-px row kept two of a 256 x 1 block's four waves busy computing on zero fill (8 x 1080p gs_sobel at 0.21
   * of the HBM peak).  Key 1: 0 / 1 / 2 force 64 x 4 / 256 x 1 / 128 x 2, anything else = by width. */
  unsigned bx = 256, by = 1;
  if (g_tune[1] == 0 || (g_tune[1] > 2 && strips <= 64)) bx = 64, by = 4;
  else if (g_tune[1] == 2 || (g_tune[1] > 2 && strips <= 128)) bx = 128, by = 2;
  c.block = dim3(bx, by);
  c.grid = dim3((strips + bx - 1) / bx, (nb + by - 1) / by, n);
  /* XCD-aware band mapping for the short-band (HBM-bound) kernels: one block per band row (w <= 4096), the band
   * count padded to a multiple of 8 (blocks past the last band return at once).  Key 18: 1 = off, 2 = always. */
  /* measured (profiles/r03g_strip_xcd_bands.log): 512 x 4K gs_blur(2) +3.1 %, gs_erode +2.6 %, gs_sobel +2.0 %,
   * 64 x 4096^2 copy +2 %, sobel -1 % (noise); gs_filter (waves_per_simd 6) -3.5 %: not for that one */
  const bool want = g_tune[18] == 2 || (g_tune[18] == 0 && waves_per_simd == 5 && nb >= 64 && topo().eight_xcds());
  if (want && c.grid.x == 1 && by == 1) {
    c.grid.y = (nb + 7u) & ~7u;
    c.xcd_flag = kStripXcdFlag;
  }
  return c;
}
/* ------------------------------------------------------------------ stencil launchers */
/* keep_cols: true = columns 0 / w-1 keep dst's bytes like the reference (the kernel re-writes
 * them unchanged); false = the caller does not care (it copies back the interior only, or zeroes
 * the frame afterwards), which saves one dword load per row in the two edge lanes. */
void launch_sobel(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n, bool keep_cols) {
  if (w < 3 || h < 3 || n == 0) return;
#ifdef GS_EXPERIMENT
  if (g_tune[22] == 1) keep_cols = false; /* probe: without the dst column reads (columns 0 / w-1 then receive junk) */
#endif
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    uint8_t *d = dst + fb * f0;
    const uint8_t *s = src + fb * f0;
    if (strip_ok(w, h, d, s) && w >= 32) {
      const int rg = strip_mode(w, s);
      const StripCfg c = strip_cfg(w, h - 2, nn, 5, 2, 6, rg);
      if (rg == 1) {
        if (keep_cols) GS_LAUNCH((k_sobel16<true, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_sobel16<false, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (rg == 2) {
        if (keep_cols) GS_LAUNCH((k_sobel16<true, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_sobel16<false, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (keep_cols) GS_LAUNCH(k_sobel16<true>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else GS_LAUNCH(k_sobel16<false>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
    } else {
      GS_LAUNCH(k_sobel_px, grid2d(w, h, nn), dim3(64, 4), 0, st, d, s, w, h, fb);
    }
  }
}

template <bool DILATE>
void launch_morph(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    uint8_t *d = dst + fb * f0;
    const uint8_t *s = src + fb * f0;
    if (strip_ok(w, h, d, s)) {
      const int rg = strip_mode(w, s);
      const StripCfg c = strip_cfg(w, h, nn, 5, 2, 4, rg);
      if (rg == 1) GS_LAUNCH((k_morph16<DILATE, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else if (rg == 2) GS_LAUNCH((k_morph16<DILATE, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else GS_LAUNCH(k_morph16<DILATE>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
    } else {
      GS_LAUNCH(k_morph_px<DILATE>, grid2d(w, h, nn), dim3(64, 4), 0, st, d, s, w, h, fb);
    }
  }
}

/* sq: the table of (p - 128)^2 (gs_match_template); only the banded form builds it -- false is returned, and nothing launched,
 * when this geometry would take the rows + columns form */
bool launch_integral(const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned *ii, bool sq) {
  if (n == 0) return true;
  hipStream_t st = ctx().s();
  const size_t fp = (size_t)w * h;
  /* the banded form takes any width >= 32 at any alignment (round 4: ragged rows, frames wider than 4096 px in column
   * chunks); key 6 = 1 or key 21 = 1: the rows + columns form */
  const bool banded = g_tune[6] != 1 && fp * 4 < 0x7fffffffull &&
                      (g_tune[21] == 1 ? (w % 16 == 0 && w <= 4096 && al16(src) && al16(ii)) : w >= 32);
  if (sq && !banded) return false;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    if (banded) {
      const bool wide = w > 4096;
      /* bands per frame: ~8 blocks per CU in flight, and for narrow frames no band taller than 32 rows (48 up to 2048 px).
       * A wave of k_integral_wave owns one band and walks it row by row; a 4K row is 16 tiles of arithmetic and 8 waves per
       * CU keep HBM busy, a 612-px row is three, and a large batch of such frames -- 8 bands of 102 rows each -- ran at
       * 2.5 TB/s (256 x 612x816 0.255 ms; with bands of 26 rows and 4-tile waves 0.21; profiles/r06h_integral_ab_tiles_and_band_height.log).
       * Every band costs 12 B per column (its sums: written, scanned, read), which is why wide frames keep tall bands:
       * 64 x 1080p with 64 bands instead of 32 measured 9 % slower. */
      unsigned nb = std::max(1u, 8u * topo().cus / nn);
      if (w <= 1024) nb = std::max(nb, (h + 31u) / 32u);
      else if (w <= 2048) nb = std::max(nb, (h + 47u) / 48u);
      nb = std::min(nb, std::max(1u, h / 8));
      if (wide) nb = std::max(nb, (h + kIntegralWideRows - 1) / kIntegralWideRows); /* a row's carry waits in LDS */
      const unsigned BH = (h + nb - 1) / nb;
      nb = (h + BH - 1) / BH;
      unsigned *cs = (unsigned *)ctx().scratch(SL_HISTP, (size_t)nn * nb * w * 4);
      const uint8_t *s = src + fp * f0;
      unsigned *o = ii + fp * f0;
      /* source planes the Infinity Cache (256 MB, shared with the table being written) cannot keep between the first pass and
       * the third: both read them with streaming loads (k_integral.h).  Same-box A/B of that policy for every batch: 16 x 4096^2
       * (256 MiB) -7 %, 64 x 4K -4 ... -12 %, 64 x 1080p (133 MB) +-0, 8 x 1080p +22 %, 32 x 720p +23 %
       * (profiles/r06i_integral_nt_loads.log, r06j_integral_ab.log).  The plain table of frames wider than 2048 px only --
       * the 16-tile waves -- to keep the instantiations few. */
      const bool nt = !sq && w > 2048 && (size_t)nn * fp >= ((size_t)192 << 20) && g_tune[6] != 8; /* key 6 = 8: default policy for every batch (A/B) */
      /* the first pass streams at that size whatever the width (256 x 720p -4 %); from 48 MiB it cost 64 - 75 MiB batches 7-12 %:
       * the third pass then misses the cache for its source (profiles/r06aa_integral_first_pass_nt.log) */
      const bool nt_first = !sq && (size_t)nn * fp >= ((size_t)192 << 20) && g_tune[6] != 8;
      if (sq) GS_LAUNCH(k_integral_colsum<true>, dim3((w + 4095) / 4096, nb, nn), dim3(256), 0, st, s, w, h, BH, nb, cs);
      else if (nt || nt_first) GS_LAUNCH((k_integral_colsum<false, true>), dim3((w + 4095) / 4096, nb, nn), dim3(256), 0, st, s, w, h, BH, nb, cs);
      else GS_LAUNCH(k_integral_colsum<false>, dim3((w + 4095) / 4096, nb, nn), dim3(256), 0, st, s, w, h, BH, nb, cs);
      GS_LAUNCH(k_integral_colbase, dim3((w + 63) / 64, nn), dim3(64, 16), 0, st, cs, w, nb);
      const dim3 gw(1, (nb + 3) / 4, nn);
      const bool rg = (w & 3u) != 0u;
      if (sq) { /* 16 tiles whatever the width: four instantiations instead of six */
        if (wide && rg) GS_LAUNCH((k_integral_wave<16, true, true, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
        else if (wide) GS_LAUNCH((k_integral_wave<16, false, true, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
        else if (rg) GS_LAUNCH((k_integral_wave<16, true, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
        else GS_LAUNCH((k_integral_wave<16, false, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      } else if (g_tune[6] == 2 && !rg && !wide && w % 16 == 0) /* the block-per-band form (one barrier per row), kept for comparison */
        GS_LAUNCH(k_integral_band, dim3(1, nb, nn), dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (nt && wide && rg) GS_LAUNCH((k_integral_wave<16, true, true, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (nt && wide) GS_LAUNCH((k_integral_wave<16, false, true, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (nt && rg) GS_LAUNCH((k_integral_wave<16, true, false, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (nt) GS_LAUNCH((k_integral_wave<16, false, false, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (wide && rg) GS_LAUNCH((k_integral_wave<16, true, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (wide) GS_LAUNCH((k_integral_wave<16, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      /* tiles of 256 px per row: the smallest of 2 / 4 / 8 / 16 that spans it (a tile past the row's end still costs its wave scan) */
      else if (w <= 512 && rg) GS_LAUNCH((k_integral_wave<2, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 512) GS_LAUNCH(k_integral_wave<2>, gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 1024 && rg) GS_LAUNCH((k_integral_wave<4, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 1024) GS_LAUNCH(k_integral_wave<4>, gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 2048 && rg) GS_LAUNCH((k_integral_wave<8, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 2048) GS_LAUNCH(k_integral_wave<8>, gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (rg) GS_LAUNCH((k_integral_wave<16, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else GS_LAUNCH(k_integral_wave<16>, gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
    } else {
      GS_LAUNCH(k_integral_rows, dim3(h, nn), dim3(256), 0, st, src + fp * f0, w, h, ii + fp * f0);
      GS_LAUNCH(k_integral_cols, dim3((w + 255) / 256, nn), dim3(256), 0, st, ii + fp * f0, w, h);
    }
  }
  return true;
}

/* gs_blur for any radius: register strips for r = 1..3 on aligned frames, otherwise clipped box
 * sums from a scratch integral image (exact: both are u32-modular like the reference). */
template <int MODE>
void launch_box_generic(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                        unsigned radius, int c) {
  hipStream_t st = ctx().s();
  const size_t fp = (size_t)w * h;
  const unsigned r = std::min(radius, std::max(w, h)); /* larger windows clip identically */
  if (g_tune[6] != 3 && r >= 1 && r <= 127 && w <= 4096 && strip_ok(w, h, dst, src)) {
    /* sliding box sums straight from the source rows (k_box.h): 3-4 B/px instead of the ~16 of the integral-image
     * route below (64 4K frames: 2.8 ms whatever the radius; this one: r = 16 0.36 ms, r = 40 0.63 ms); the kernel's
     * u16 column sums and LDS halo hold up to r = 127 */
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      /* r <= 16: the window's raw rows stay in registers (k_box16r: 2 B/px instead of 3-4; 64 x 4K: 0.22 ms for r <= 9,
       * 0.28 up to 16, against 0.31-0.42 -- profiles/r03r_box_ring.log).  Key 6 = 4: k_box16 always. */
      /* ragged rows: the ring kernel over the whole strips (the last one only feeds its neighbour) + k_box_edge for the
       * last 16 + w % 16 columns in a launch of its own (round 5).  Round 4's RAGGED form of the ring kernels (per-pixel
       * choice between two edge divisors, shifted tail loads inside the block) was no faster than the any-radius kernel --
       * 116-152 registers at r = 4..5 instead of 84-92, one wave per SIMD from r = 12
       * (profiles/r04k_box_ring_ragged_not_kept.log: 3838 x 2160 r = 5 0.446 vs 0.434 ms, r = 16 0.73 vs 0.53).
       * Key 6 = 5: ragged rows on the any-radius kernel (rounds 2-4). */
      const bool ring = g_tune[6] != 4 && !(g_tune[6] == 5 && ragged(w)) && r <= box_ring_max() && w >= 32 && h >= 2 * r + 1 &&
                        (MODE == 0 || (c > -(1 << 30) && c < (1 << 30)));
      /* band height: the launch should be whole rounds of the blocks the chip holds (256 CUs x 4 of them, fewer for the
       * register-heavy ring kernels of r >= 8 / 10), and a band first loads 2r+1 rows it does not output -- loads and
       * adds only since the vertical-first form, ~0.3 of an output row each.  Pick the band count with the smallest
       * rounds x (T + 0.3 (2r+1)): 64 4K frames -> 16 bands of 135 rows (r = 16: 0.50 -> 0.36 ms, r = 40: 1.11 -> 0.63),
       * 8 frames -> 128 bands (r = 40: 1.0 -> 0.16 ms; profiles/r02l_box_T.log).  Round 1's "at least four window
       * heights per band" dates from a prologue that cost more than the rows it preceded. */
      const unsigned threads = w <= 1024 ? 64u : w <= 2048 ? 128u : 256u; /* a thread owns 16 px of the row */
      const unsigned slots = topo().cus * box_blocks_per_cu(MODE, ring ? r : 0u, threads);
      unsigned T = h;
      if (g_tune[0] > 0) {
        T = (unsigned)g_tune[0];
      } else {
        /* once the launch is several full rounds, more bands only add prologues (cost ~ nn h / slots + nn nbc k / slots
         * grows with nbc), so the search stops at 4 rounds' worth of blocks: at most 4096 candidates however tall
         * the image is; the last answer is kept per (h, nn, r, slots) */
        static thread_local struct { unsigned h, nn, r, slots, T; } memo = {0, 0, 0, 0, 0};
        if (memo.h == h && memo.nn == nn && memo.r == r && memo.slots == slots) {
          T = memo.T;
        } else {
          double best = 1e30;
          const unsigned nbc_max = std::max(1u, std::min(h / 8u, std::max(1u, 4u * slots / nn)));
          for (unsigned nbc = 1; nbc <= nbc_max; nbc++) {
            const unsigned t = (h + nbc - 1) / nbc;
            const double rounds = (double)(((unsigned long long)nn * ((h + t - 1) / t) + slots - 1) / slots);
            const double cost = rounds * ((double)t + (ring ? 0.12 : 0.3) * (2.0 * r + 1.0)); /* ring: 16 VALU per start-up row */
            if (cost < best - 1e-9) best = cost, T = t;
          }
          memo = {h, nn, r, slots, T};
        }
      }
      const unsigned nb = (h + T - 1) / T;
      if (ring && ragged(w)) {
        /* about four waves per SIMD, and bands at least two windows tall (a band starts with 2 r + 1 rows of loads) */
        const unsigned want = std::max(1u, 4u * 4u * topo().cus / nn);
        const unsigned Te = g_tune[0] > 0 ? (unsigned)g_tune[0] : std::max((h + want - 1) / want, std::min(h, 2u * (2u * r + 1u)));
        const dim3 ge(1, (h + Te - 1) / Te, nn);
        /* The edge launch is a chain of memory latencies (15-30 us however little it computes, growing with the 2 r + 1
         * start-up rows) and touches other columns of dst than the body.  For r >= 12 on large batches it goes to the side
         * stream, ordered behind everything the caller's stream holds so far, and the caller's stream waits for it after the
         * body: it hides under the body launch instead of following it (64 x 3838x2160: r = 16 0.307 -> 0.296 ms, adaptive
         * r = 15 0.308 -> 0.292; r = 5 and 9 are 2 % better off in sequence -- profiles/r05l_box_ragged_buffer_ops.log).
         * Key 6 = 6: always on the caller's stream, 7: always on the side stream. */
        bool side = false;
#ifndef GS_EMU
        side = g_tune[6] == 7 || (g_tune[6] != 6 && r >= 12u && (size_t)nn * fp >= ((size_t)16 << 20));
        if (side) {
          Ctx &cx = ctx();
          cx.ensure_side();
          GS_HIP(hipEventRecord(cx.ev_chunk[0], st));
          GS_HIP(hipStreamWaitEvent(cx.side, cx.ev_chunk[0], 0));
          launch_box_edge(MODE, ge, cx.side, dst + fp * f0, src + fp * f0, w, h, Te, fp, r, c);
          GS_HIP(hipEventRecord(cx.ev_join, cx.side));
        }
#endif
        launch_box(MODE, r, dim3(1, nb, nn), threads, st, dst + fp * f0, src + fp * f0, w, h, T, fp, r, c);
#ifndef GS_EMU
        if (side) GS_HIP(hipStreamWaitEvent(st, ctx().ev_join, 0));
#endif
        if (!side) launch_box_edge(MODE, ge, st, dst + fp * f0, src + fp * f0, w, h, Te, fp, r, c);
      } else {
        launch_box(MODE, ring ? r : 0u, dim3(1, nb, nn), threads, st, dst + fp * f0, src + fp * f0, w, h, T, fp, r, c);
      }
    }
    return;
  }
  const unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(n, (256u << 20) / (fp * 4) + 1));
  unsigned *ii = (unsigned *)ctx().scratch(SL_II, fp * 4 * group);
  for (unsigned f0 = 0; f0 < n; f0 += group) {
    const unsigned nn = std::min(group, n - f0);
    launch_integral(src + fp * f0, w, h, nn, ii);
    GS_LAUNCH(k_box_px<MODE>, grid2d(w, h, nn), dim3(64, 4), 0, st, dst + fp * f0, src + fp * f0,
              (const unsigned *)ii, w, h, r, c, fp);
  }
}

void launch_blur(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                 unsigned radius) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  if (radius >= 1 && radius <= 3 && strip_ok(w, h, dst, src) && h > 2 * radius && w > 2 * radius) {
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      uint8_t *d = dst + fb * f0;
      const uint8_t *s = src + fb * f0;
      const int rg = strip_mode(w, s);
      const StripCfg c = strip_cfg(w, h, nn, 5, radius, radius == 1 ? 4 : radius == 2 ? 6 : 12, rg);
      if (rg == 1) {
        if (radius == 1) GS_LAUNCH((k_blur16<1, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else if (radius == 2) GS_LAUNCH((k_blur16<2, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_blur16<3, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (rg == 2) {
        if (radius == 1) GS_LAUNCH((k_blur16<1, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else if (radius == 2) GS_LAUNCH((k_blur16<2, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_blur16<3, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (radius == 1) GS_LAUNCH(k_blur16<1>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else if (radius == 2) GS_LAUNCH(k_blur16<2>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else GS_LAUNCH(k_blur16<3>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      /* the 2*radius vertically clipped rows of each frame get their true divisors */
      GS_LAUNCH(k_blur_edge_rows, dim3((w + 255) / 256, 2 * radius, nn), dim3(256), 0, st, d, s, w, h,
                (int)radius, fb);
    }
    return;
  }
  if (radius == 0) { /* 1x1 window: identity (sum/1) */
    if (dst != src) GS_HIP(hipMemcpyAsync(dst, src, fb * n, hipMemcpyDeviceToDevice, st));
    return;
  }
  launch_box_generic<0>(dst, src, w, h, n, radius, 0);
}

/* ------------------------------------------------------------------ histogram / otsu / threshold */
/* blocks per frame of k_hist_partial.  A block pays for zeroing and folding its 32 KB of LDS counters and for
 * filling its load queue, so it should run ~64 trips of 16 B per lane (512 4K frames: 4.9 Tpx/s at 31 trips,
 * 5.8 at 63; profiles/r02i_hist.log, r02i_hist_trips.log); a handful of frames is spread over the CUs
 * (256 CUs x 5 resident blocks) down to 16 trips per block, at most 256 blocks per frame for k_hist_reduce. */
constexpr unsigned kHistThreads = 256;
unsigned hist_threads() { return g_tune[10] >= 1000 ? (unsigned)(g_tune[10] / 1000) * 256u : kHistThreads; } /* experiments: key 10 = 1000 * (threads / 256) + trips */
unsigned hist_bpf(size_t frame_bytes, unsigned n) {
  if (g_tune[11] > 0) return (unsigned)g_tune[11];
  const size_t bt = hist_threads();
  const size_t chunks = frame_bytes / 16 + 1, trips = g_tune[10] % 1000 > 0 ? (size_t)(g_tune[10] % 1000) : 64 * 256 / bt;
  const size_t by_size = (chunks + bt * trips - 1) / (bt * trips);
  const size_t by_fill = std::min<size_t>(std::min<size_t>((5u * topo().cus + n - 1) / n, chunks / (bt * 16)), 256); /* 5 resident blocks per CU */
  return (unsigned)std::max<size_t>(1, std::min<size_t>(std::max(by_size, by_fill), 2048));
}
void launch_hist_partial(dim3 grid, hipStream_t st, const uint8_t *img, size_t frame_bytes, unsigned *partial) {
  /* more bytes than the Infinity Cache keeps (192 MiB and up): streaming loads (k_pointwise.h); key 10 = -1: never */
  const bool nt = (size_t)grid.y * grid.z * frame_bytes >= ((size_t)192 << 20) && g_tune[10] != -1;
  switch (hist_threads()) {
    case 512: GS_LAUNCH(k_hist_partial<512>, grid, dim3(512), 0, st, img, frame_bytes, partial); break;
    case 1024: GS_LAUNCH(k_hist_partial<1024>, grid, dim3(1024), 0, st, img, frame_bytes, partial); break;
    default:
      if (nt) GS_LAUNCH((k_hist_partial<256, true>), grid, dim3(256), 0, st, img, frame_bytes, partial);
      else GS_LAUNCH(k_hist_partial<256>, grid, dim3(256), 0, st, img, frame_bytes, partial);
  }
}
void launch_histogram(const uint8_t *img, size_t frame_bytes, unsigned n, unsigned *hist) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  /* k_hist_partial addresses a frame with 32-bit offsets: count a huge image in pieces (key 12: piece size in
   * bytes, so that tests reach this path with small images) */
  const size_t piece_bytes = g_tune[12] > 0 ? (size_t)g_tune[12] : kHistMaxFrame;
  if (frame_bytes > piece_bytes) {
    const unsigned pieces = (unsigned)(frame_bytes / piece_bytes);
    const size_t rest = frame_bytes - (size_t)pieces * piece_bytes;
    const unsigned bpf = hist_bpf(piece_bytes, pieces), bpr = rest ? hist_bpf(rest, 1) : 0;
    unsigned *partial = (unsigned *)ctx().scratch(SL_HISTP, ((size_t)pieces * bpf + bpr) * 256 * 4);
    for (unsigned f = 0; f < n; f++) {
      const uint8_t *p = img + frame_bytes * f;
      launch_hist_partial(dim3(bpf, pieces), st, p, piece_bytes, partial);
      if (rest)
        launch_hist_partial(dim3(bpr, 1), st, p + (size_t)pieces * piece_bytes, rest, partial + (size_t)pieces * bpf * 256);
      GS_LAUNCH(k_hist_reduce, dim3(1), dim3(256), 0, st, (const unsigned *)partial, pieces * bpf + bpr,
                hist + (size_t)f * 256, 0u);
    }
    return;
  }
  const unsigned bpf = hist_bpf(frame_bytes, n);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    unsigned *partial = (unsigned *)ctx().scratch(SL_HISTP, (size_t)nn * bpf * 256 * 4);
    launch_hist_partial(dim3(bpf, nn), st, img + frame_bytes * f0, frame_bytes, partial);
    GS_LAUNCH(k_hist_reduce, dim3(nn), dim3(256), 0, st, (const unsigned *)partial, bpf,
              hist + (size_t)f0 * 256, 0u);
  }
}
void launch_threshold(uint8_t *img, size_t frame_bytes, unsigned n, const uint8_t *thr_dev,
                      unsigned thr_const, hipStream_t on = nullptr) {
  if (n == 0) return;
  hipStream_t st = on ? on : ctx().s();
  const size_t chunks = frame_bytes / 16 + 2;
  const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((chunks + 255) / 256, 2048));
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_threshold, dim3(bx, nn), dim3(256), 0, st, img + frame_bytes * f0, frame_bytes,
              thr_dev ? thr_dev + f0 : nullptr, thr_const);
  }
}
void launch_otsu(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist,
                 uint8_t *thr) {
  launch_histogram(img, (size_t)(w * h), n, hist);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_otsu, dim3(nn), dim3(256), 0, ctx().s(), hist + (size_t)f0 * 256, w * h, thr + f0,
              (const unsigned *)nullptr, 0u, 0u);
  }
}

void synth_jump_table(SynthJump &J) {
  auto step = [](uint32_t x) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
  };
  for (int b = 0; b < 32; b++) J.col[0][b] = step(1u << b);
  for (int k = 1; k < 32; k++)
    for (int b = 0; b < 32; b++) {
      uint32_t v = J.col[k - 1][b], r = 0;
      for (int q = 0; q < 32; q++)
        if ((v >> q) & 1u) r ^= J.col[k - 1][q];
      J.col[k][b] = r;
    }
}

void launch_integral_pad(dim3 grid, hipStream_t st, const unsigned *ii, unsigned w, unsigned h, unsigned *padded) {
  GS_LAUNCH(k_integral_pad, grid, dim3(64, 4), 0, st, ii, w, h, padded);
}

}  // namespace gsi

extern "C" {

#ifdef GS_EXPERIMENT
void gsh_probe_strip_copy(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w >= 32);
  const StripCfg c = strip_cfg(w, h, n, 5, 2, g_tune[0] > 0 ? (unsigned)g_tune[0] : 8u, ragged(w) ? 1 : 0);
  if (g_tune[23] == 1) GS_LAUNCH((k_strip_copy<0, true>), c.grid, c.block, 0, ctx().s(), dst, src, w, h, c.T, (size_t)w * h | c.xcd_flag);
  else if (ragged(w)) GS_LAUNCH(k_strip_copy<1>, c.grid, c.block, 0, ctx().s(), dst, src, w, h, c.T, (size_t)w * h | c.xcd_flag);
  else GS_LAUNCH(k_strip_copy<0>, c.grid, c.block, 0, ctx().s(), dst, src, w, h, c.T, (size_t)w * h | c.xcd_flag);
}
#endif
/* ---------------------------------------------------------------- batch: stencils */
void gsh_blur_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                    unsigned radius) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_blur(dst, src, w, h, n, radius);
}
void gsh_sobel_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_sobel(dst, src, w, h, n);
}
void gsh_erode_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_morph<false>(dst, src, w, h, n);
}
void gsh_dilate_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_morph<true>(dst, src, w, h, n);
}

/* ---------------------------------------------------------------- batch: histogram etc. */
void gsh_histogram_batch(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist) {
  GS_ASSERT(img && hist && w > 0 && h > 0);
  launch_histogram(img, (size_t)(w * h), n, hist); /* 32-bit product like ref :202 */
}
void gsh_otsu_batch(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist_scratch,
                    uint8_t *thr) {
  GS_ASSERT(img && hist_scratch && thr && w > 0 && h > 0);
  launch_otsu(img, w, h, n, hist_scratch, thr);
}
void gsh_threshold_batch(uint8_t *img, unsigned w, unsigned h, unsigned n, uint8_t thresh) {
  GS_ASSERT(img && w > 0 && h > 0);
  launch_threshold(img, (size_t)(w * h), n, nullptr, thresh);
}
void gsh_threshold_batch_dev(uint8_t *img, unsigned w, unsigned h, unsigned n, const uint8_t *thr) {
  GS_ASSERT(img && thr && w > 0 && h > 0);
  launch_threshold(img, (size_t)(w * h), n, thr, 0);
}
/* gs_blur(radius) then gs_sobel into a zeroed image, per frame, in one pass (the fused kernel of
 * the pipeline without the Otsu / threshold half) */
void gsh_blur_sobel_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                          unsigned radius) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  if (n == 0) return;
  const size_t fb = (size_t)w * h;
  hipStream_t st = ctx().s();
  if (g_tune[3] == 0 && radius >= 1 && radius <= 3 && strip_ok(w, h, dst, src) && w >= 32 && h >= 3 &&
      h > 2 * radius) {
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      const int rg = strip_mode(w, src + fb * f0);
      const StripCfg c = strip_cfg(w, h - 2, nn, 3, 2, 8, rg);
      launch_blur_sobel(radius, dim3(c.grid.x, c.grid.y, nn), c.block, st, dst + fb * f0, src + fb * f0, w, h,
                        c.T, fb, rg);
      GS_LAUNCH(k_zero_frame, dim3((2 * w + 2 * h + 255) / 256, nn), dim3(256), 0, st, dst + fb * f0, w, h, fb);
    }
    return;
  }
  uint8_t *t = (uint8_t *)ctx().scratch(SL_AUX, fb * n);
  launch_blur(t, src, w, h, n, radius);
  GS_HIP(hipMemsetAsync(dst, 0, fb * n, st));
  if (w >= 3 && h >= 3) launch_sobel(dst, t, w, h, n, true);
}
void gsh_edge_pipeline_batch(uint8_t *dst, uint8_t *tmp, const uint8_t *src, unsigned w, unsigned h,
                             unsigned n, unsigned radius, unsigned *hist_scratch, uint8_t *thr) {
  GS_ASSERT(dst && src && hist_scratch && thr && w > 0 && h > 0);
  const size_t fb = (size_t)w * h;
  hipStream_t st = ctx().s();
  auto zero_frame = [&]() { /* gs_sobel ran "into a zeroed image": only its 1-px frame is left */
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ)
      GS_LAUNCH(k_zero_frame, dim3((2 * w + 2 * h + 255) / 256, std::min(kMaxZ, n - f0)), dim3(256), 0,
                st, dst + fb * f0, w, h, fb);
  };
  if (!tmp && g_tune[3] == 0 && radius >= 1 && radius <= 3 && strip_ok16(w, h, dst, src) && w >= 32 &&
      h >= 3 && h > 2 * radius) { /* every window is clipped on at most one side per axis */
    /* fused: the blurred image only ever exists in registers (1 R + 1 W per pixel).
     * The fused kernel is VALU-bound (~40 % of HBM peak) and the passes after it HBM-bound with
     * an idle VALU.  A batch larger than kChunkFrames is cut into chunks: the fused kernels run
     * back to back on the caller's stream, each chunk's histogram reduce / Otsu / frame zeroing /
     * threshold pass on a side stream behind one event, i.e. under the next chunk's fused kernel.
     * gsh_tune key 5: frames per chunk (0 = default, negative = never split). */
    const int tune_chunk = g_tune[5];
    const unsigned per = tune_chunk < 0 ? n : tune_chunk > 0 ? (unsigned)tune_chunk : kChunkFrames;
    /* chunk sizes: `per` frames each.  (Tapering the tail -- 16, 8, 8 -- so that less of the last
     * threshold pass is exposed measured slower: small fused launches cost more than they hide.) */
    std::vector<unsigned> sizes;
    for (unsigned rem = n; rem; rem -= std::min(rem, per)) sizes.push_back(std::min(rem, per));
    /* band height per chunk size (a smaller chunk needs more bands to fill the chip) */
    auto cfg_for = [&](unsigned nn) { return strip_cfg(w, h - 2, std::min(kMaxZ, nn), 3); /* <= 168 VGPRs: 3 waves per SIMD */ };
    unsigned bpf_max = 0;
    for (unsigned nn : sizes) {
      const StripCfg c = cfg_for(nn);
      bpf_max = std::max(bpf_max, c.grid.x * c.grid.y);
    }
    unsigned *partial = (unsigned *)ctx().scratch(SL_HISTP, (size_t)n * bpf_max * 256 * 4);
    auto run_fused = [&](hipStream_t on, unsigned f0, unsigned nn) {
      const StripCfg c = cfg_for(nn);
      for (unsigned g0 = f0; g0 < f0 + nn; g0 += kMaxZ) {
        const unsigned m = std::min(kMaxZ, f0 + nn - g0);
        const dim3 grid(c.grid.x, c.grid.y, m);
        uint8_t *d = dst + fb * g0;
        const uint8_t *sp = src + fb * g0;
        unsigned *pp = partial + (size_t)g0 * bpf_max * 256;
#ifndef GS_EMU
        ctx().prof_mark(0, on);
#endif
        launch_blur_sobel_hist(radius, grid, c.block, on, d, sp, w, h, c.T, fb, pp);
#ifndef GS_EMU
        ctx().prof_mark(1, on);
#endif
      }
    };
    auto run_rest = [&](hipStream_t on, unsigned f0, unsigned nn) {
      const StripCfg c = cfg_for(nn);
      const unsigned bpf = c.grid.x * c.grid.y;
      for (unsigned g0 = f0; g0 < f0 + nn; g0 += kMaxZ) {
        const unsigned m = std::min(kMaxZ, f0 + nn - g0);
        const unsigned *pp = partial + (size_t)g0 * bpf_max * 256;
        /* the 2w + 2(h-2) frame pixels are 0 in the result and were not counted by the kernel; the threshold pass
         * below waits for this launch only (k_otsu folds the blocks' partial histograms itself) */
        GS_LAUNCH(k_otsu, dim3(m), dim3(256), 0, on, hist_scratch + (size_t)g0 * 256, w * h, thr + g0, pp, bpf,
                  2 * w + 2 * (h - 2));
      }
      launch_threshold(dst + fb * f0, fb, nn, thr + f0, 0, on);
      /* gs_sobel ran "into a zeroed image": only its 1-px frame is left to zero.  AFTER the threshold pass (which
       * turns whatever the fused kernel left there into 0 / 255): the frame is then 0 either way, and the
       * threshold pass need not wait for this launch. */
      for (unsigned g0 = f0; g0 < f0 + nn; g0 += kMaxZ) {
        const unsigned m = std::min(kMaxZ, f0 + nn - g0);
        GS_LAUNCH(k_zero_frame, dim3((2 * w + 2 * h + 255) / 256, m), dim3(256), 0, on, dst + fb * g0, w, h, fb);
      }
    };
#ifdef GS_EMU
    const bool split = false;
#else
    const bool split = sizes.size() > 1 && sizes.size() <= kMaxChunks;
#endif
    if (!split) {
      run_fused(st, 0, n);
      run_rest(st, 0, n);
      return;
    }
#ifndef GS_EMU
    Ctx &cx = ctx();
    cx.ensure_side();
    unsigned f0 = 0;
    for (size_t i = 0; i < sizes.size(); i++) {
      const unsigned nn = sizes[i];
      run_fused(st, f0, nn);
      if (i + 1 == sizes.size()) { /* last chunk: nothing left to hide it under; rejoin the caller's stream */
        GS_HIP(hipEventRecord(cx.ev_join, cx.side));
        GS_HIP(hipStreamWaitEvent(st, cx.ev_join, 0));
        run_rest(st, f0, nn);
      } else {
        GS_HIP(hipEventRecord(cx.ev_chunk[i], st));
        GS_HIP(hipStreamWaitEvent(cx.side, cx.ev_chunk[i], 0));
        run_rest(cx.side, f0, nn);
      }
      f0 += nn;
    }
#endif
    return;
  }
  if (!tmp && g_tune[3] == 0 && radius >= 1 && radius <= 3 && strip_ok(w, h, dst, src) && w >= 32 && h >= 3 &&
      h > 2 * radius) {
    /* ragged rows or frames at odd addresses: blur + sobel still in one pass (the fused kernel without its histogram
     * half), the histogram as a pass of its own: 5 B/px moved instead of 4, against 9 for the separate calls */
    gsh_blur_sobel_batch(dst, src, w, h, n, radius);
    launch_otsu(dst, w, h, n, hist_scratch, thr);
    launch_threshold(dst, fb, n, thr, 0);
    return;
  }
  uint8_t *t = tmp ? tmp : (uint8_t *)ctx().scratch(SL_AUX, fb * n);
  launch_blur(t, src, w, h, n, radius);
  /* sobel never writes its 1-px frame (ref :308-309); config 2 runs it into a zeroed image, so
   * only that frame needs zeroing (the interior is overwritten) -- after the sobel launch,
   * which then need not preserve columns 0 / w-1 */
  if (w < 3 || h < 3) GS_HIP(hipMemsetAsync(dst, 0, fb * n, st));
  launch_sobel(dst, t, w, h, n, false);
  if (w >= 3 && h >= 3) zero_frame();
  launch_otsu(dst, w, h, n, hist_scratch, thr);
  launch_threshold(dst, fb, n, thr, 0);
}

/* ---------------------------------------------------------------- batch: integral + LBP */
void gsh_integral_batch(const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned *ii) {
  GS_ASSERT(src && ii && w > 0 && h > 0);
  launch_integral(src, w, h, n, ii);
}

/* ---------------------------------------------------------------- batch: "next" rows */
void gsh_adaptive_threshold_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                                  unsigned n, unsigned radius, int c) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  if (n) launch_box_generic<1>(dst, src, w, h, n, radius, c);
}
void gsh_filter_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                      const int8_t *kernel_host, unsigned kw, unsigned kh, unsigned norm) {
  GS_ASSERT(dst && src && kernel_host && w > 0 && h > 0 && kw > 0 && kh > 0 && norm > 0);
  hipStream_t st = ctx().s();
  int8_t *dk = (int8_t *)ctx().scratch(SL_TAB, (size_t)kw * kh);
  GS_HIP(hipMemcpyAsync(dk, kernel_host, (size_t)kw * kh, hipMemcpyHostToDevice, st));
  ctx().sync(); /* kernel_host may be a temporary */
  const size_t fb = (size_t)w * h;
  /* strip kernel: 3x3, every partial sum within int16 (sum |k| <= 128), norm <= 256 */
  unsigned abs_sum = 0;
  for (unsigned i = 0; i < kw * kh; i++) abs_sum += (unsigned)std::abs((int)kernel_host[i]);
  if (kw == 3 && kh == 3 && abs_sum <= 128 && norm <= 256 && strip_ok(w, h, dst, src) && w >= 32) {
    FilterK fk;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) fk.k[r][c] = ((uint32_t)(uint16_t)(int16_t)kernel_host[r * 3 + c]) * 0x10001u;
    fk.mul = (0x1000000u + norm - 1u) / norm;
    fk.cap = std::min(255u * norm, 32767u) * 0x10001u;
    fk.neg_is_255 = norm > 1 ? 0xffffffffu : 0u;
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      const int rg = strip_mode(w, src + fb * f0);
      const StripCfg c = strip_cfg(w, h, nn, 6, 2, 8, rg);
      uint8_t *d = dst + fb * f0;
      const uint8_t *s = src + fb * f0;
      const size_t fa = fb | c.xcd_flag;
      if (norm == 1) { /* multiplier 2^24: a shift (k_stencil.h) */
        if (rg == 1) GS_LAUNCH((k_filter16<1, true>), c.grid, c.block, 0, st, d, s, w, h, c.T, fa, fk);
        else if (rg == 2) GS_LAUNCH((k_filter16<2, true>), c.grid, c.block, 0, st, d, s, w, h, c.T, fa, fk);
        else GS_LAUNCH((k_filter16<0, true>), c.grid, c.block, 0, st, d, s, w, h, c.T, fa, fk);
      } else if (rg == 1) GS_LAUNCH(k_filter16<1>, c.grid, c.block, 0, st, d, s, w, h, c.T, fa, fk);
      else if (rg == 2) GS_LAUNCH(k_filter16<2>, c.grid, c.block, 0, st, d, s, w, h, c.T, fa, fk);
      else GS_LAUNCH(k_filter16<0>, c.grid, c.block, 0, st, d, s, w, h, c.T, fa, fk);
    }
    return;
  }
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_filter_px, grid2d(w, h, nn), dim3(64, 4), 0, st, dst + fb * f0, src + fb * f0, w, h,
              fb, (const int8_t *)dk, kw, kh, norm);
  }
}
void gsh_downsample_batch(uint8_t *dst, const uint8_t *src, unsigned sw, unsigned sh, unsigned n) {
  GS_ASSERT(dst && src && sw > 1 && sh > 1);
  hipStream_t st = ctx().s();
  const size_t sfb = (size_t)sw * sh, dfb = (size_t)(sw / 2) * (sh / 2);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    if (g_tune[21] == 1 ? (sw % 16 == 0 && al16(src) && al16(dst) && sfb % 16 == 0 && dfb % 8 == 0) : sw >= 16)
      GS_LAUNCH(k_downsample8, dim3(((sw / 2 + 7) / 8 + 63) / 64, (sh / 2 + 3) / 4, nn), dim3(64, 4), 0, st,
                dst + dfb * f0, src + sfb * f0, sw, sh);
    else
      GS_LAUNCH(k_downsample_px, grid2d(sw / 2, sh / 2, nn), dim3(64, 4), 0, st, dst + dfb * f0,
                src + sfb * f0, sw, sh);
  }
}

/* ---------------------------------------------------------------- synthetic frames, checksums */
void gsh_synth_batch(uint8_t *dst, unsigned w, unsigned h, unsigned n, uint32_t seed0) {
  GS_ASSERT(dst && w > 0 && h > 0);
  if (!n) return;
  hipStream_t st = ctx().s();
  bool &jump_ready = ctx().jump_ready; /* cleared when the scratch is released (shutdown / device switch) */
  SynthJump *dj = (SynthJump *)ctx().scratch(SL_JUMP, sizeof(SynthJump));
  if (!jump_ready) {
    SynthJump J;
    synth_jump_table(J);
    GS_HIP(hipMemcpyAsync(dj, &J, sizeof J, hipMemcpyHostToDevice, st));
    ctx().sync();
    jump_ready = true;
  }
  const unsigned nlev = ((w + 31) / 32) * ((h + 31) / 32);
  const size_t npx = (size_t)w * h;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    uint8_t *lev = (uint8_t *)ctx().scratch(SL_LEV, (size_t)nlev * nn);
    const unsigned per = 256 * kSynthRun;
    GS_LAUNCH(k_synth_levels, dim3((nlev + per - 1) / per, nn), dim3(256), 0, st, lev, nlev,
              seed0 + f0, (const SynthJump *)dj);
    GS_LAUNCH(k_synth_pixels, dim3((unsigned)((npx + per - 1) / per), nn), dim3(256), 0, st,
              dst + npx * f0, (const uint8_t *)lev, w, h, seed0 + f0, (const SynthJump *)dj);
  }
}
void gsh_checksum_batch(const uint8_t *img, size_t frame_bytes, unsigned n, uint64_t *sums) {
  GS_ASSERT(img && sums && frame_bytes > 0);
  if (!n) return;
  hipStream_t st = ctx().s();
  GS_HIP(hipMemsetAsync(sums, 0, (size_t)n * 8, st));
  const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((frame_bytes + 4095) / 4096, 256));
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_checksum, dim3(bx, nn), dim3(256), 0, st, img + frame_bytes * f0, frame_bytes,
              (unsigned long long *)sums + f0);
  }
}

/* =====================================================================================
 *              drop-in functions: the reference's own names and signatures
 * ===================================================================================== */

/* dst written in full by the kernel: upload src, run, download dst */
static void unary_full(struct gs_image dst, struct gs_image src,
                       void (*run)(uint8_t *, const uint8_t *, unsigned, unsigned, unsigned,
                                   unsigned, int),
                       unsigned p0, int p1) {
  const size_t nb = (size_t)src.w * src.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, nb) : dst.data;
  run(d, s, src.w, src.h, 1, p0, p1);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, nb, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

void gs_blur(struct gs_image dst, struct gs_image src, unsigned radius) { /* ref :268 */
  GS_ASSERT(GS_VALID(src) && GS_VALID(dst) && dst.w == src.w && dst.h == src.h);
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned r, int) { launch_blur(d, s, w, h, n, r); }, radius, 0);
}
void gs_erode(struct gs_image dst, struct gs_image src) { /* ref :303 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h);
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned, int) { launch_morph<false>(d, s, w, h, n); }, 0, 0);
}
void gs_dilate(struct gs_image dst, struct gs_image src) { /* ref :304 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h);
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned, int) { launch_morph<true>(d, s, w, h, n); }, 0, 0);
}
void gs_adaptive_threshold(struct gs_image dst, struct gs_image src, unsigned radius, int c) {
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h); /* ref :232 */
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned r, int cc) { launch_box_generic<1>(d, s, w, h, n, r, cc); },
             radius, c);
}

void gs_sobel(struct gs_image dst, struct gs_image src) { /* ref :306 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h);
  const unsigned w = src.w, h = src.h;
  if (w < 3 || h < 3) return; /* reference loops are empty */
  const size_t nb = (size_t)w * h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, nb) : dst.data;
  if (dhost) {
    /* The 1-px frame of dst is never written (ref :308-309).  Rows 0 / h-1 are simply not copied
     * back; columns 0 / w-1 of the other rows are planted into the device copy (2h bytes up) so
     * rows 1..h-2 can come back as ONE contiguous copy (a pitched interior copy is 3x slower). */
    std::vector<uint8_t> cols(2 * (size_t)h);
    for (unsigned y = 1; y + 1 < h; y++) cols[2 * y] = dst.data[(size_t)y * w], cols[2 * y + 1] = dst.data[(size_t)y * w + w - 1];
    uint8_t *dc = (uint8_t *)ctx().scratch(SL_AUX2, 2 * (size_t)h);
    GS_HIP(hipMemcpyAsync(dc, cols.data(), 2 * (size_t)h, hipMemcpyHostToDevice, ctx().s()));
    GS_LAUNCH(k_put_cols, dim3((2 * h + 255) / 256), dim3(256), 0, ctx().s(), d, (const uint8_t *)dc, w, h);
    ctx().sync(); /* cols is a local */
  }
  launch_sobel(d, s, w, h, 1, true);
  if (dhost)
    GS_HIP(hipMemcpyAsync(dst.data + w, d + w, (size_t)w * (h - 2), hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

void gs_filter(struct gs_image dst, struct gs_image src, struct gs_image kernel, unsigned norm) {
  GS_ASSERT(GS_VALID(src) && GS_VALID(dst) && dst.w == src.w && dst.h == src.h && norm > 0);
  const size_t nb = (size_t)src.w * src.h, kb = (size_t)kernel.w * kernel.h;
  /* ref :260-261: the kernel is read through gs_get, so an invalid kernel contributes nothing */
  std::vector<int8_t> k(std::max<size_t>(1, kb), 0);
  unsigned kw = kernel.w, kh = kernel.h;
  if (GS_VALID(kernel)) {
    if (is_dev(kernel.data)) gsh_download(k.data(), kernel.data, kb);
    else memcpy(k.data(), kernel.data, kb);
  } else {
    kw = kh = 1; /* empty loop in the reference: sum = 0 */
    k[0] = 0;
  }
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, nb) : dst.data;
  gsh_filter_batch(d, s, src.w, src.h, 1, k.data(), kw, kh, norm);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, nb, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

void gs_downsample(struct gs_image dst, struct gs_image src) { /* ref :189 */
  GS_ASSERT(GS_VALID(src) && GS_VALID(dst) && dst.w == src.w / 2 && dst.h == src.h / 2);
  const size_t nb = (size_t)src.w * src.h, db = (size_t)dst.w * dst.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, db) : dst.data;
  gsh_downsample_batch(d, s, src.w, src.h, 1);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, db, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

/* ---- geometry + template matching ------------------------------------------ */
void gs_crop(struct gs_image dst, struct gs_image src, struct gs_rect roi) { /* ref :154 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && roi.x + roi.w <= src.w && roi.y + roi.h <= src.h &&
            dst.w == roi.w && dst.h == roi.h);
  const size_t nb = (size_t)src.w * src.h, db = (size_t)dst.w * dst.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, db) : dst.data;
  GS_LAUNCH(k_crop, dim3((roi.w + 63) / 64, (roi.h + 3) / 4), dim3(64, 4), 0, ctx().s(), d, dst.w, dst.h, s,
            src.w, src.h, roi.x, roi.y, roi.w, roi.h);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, db, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}
void gs_copy(struct gs_image dst, struct gs_image src) { /* ref :160 */
  const struct gs_rect all = {0, 0, src.w, src.h};
  gs_crop(dst, src, all);
}
static void resize_common(struct gs_image dst, struct gs_image src, bool nearest) {
  const size_t nb = (size_t)src.w * src.h, db = (size_t)dst.w * dst.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, db) : dst.data;
  const dim3 g((dst.w + 63) / 64, (dst.h + 3) / 4);
  if (nearest) GS_LAUNCH(k_resize_nn, g, dim3(64, 4), 0, ctx().s(), d, dst.w, dst.h, s, src.w, src.h);
  else GS_LAUNCH(k_resize, g, dim3(64, 4), 0, ctx().s(), d, dst.w, dst.h, s, src.w, src.h);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, db, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}
void gs_resize_nn(struct gs_image dst, struct gs_image src) { /* ref :164 (asserts nothing there) */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src));
  resize_common(dst, src, true);
}
void gs_resize(struct gs_image dst, struct gs_image src) { /* ref :171 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src));
  resize_common(dst, src, false);
}
void gs_match_template(struct gs_image img, struct gs_image tmpl, struct gs_image result) { /* ref :705 */
  GS_ASSERT(GS_VALID(img) && GS_VALID(tmpl) && GS_VALID(result));
  GS_ASSERT(img.w >= tmpl.w && img.h >= tmpl.h);
  GS_ASSERT(result.w == img.w - tmpl.w + 1 && result.h == img.h - tmpl.h + 1);
  const size_t ib = (size_t)img.w * img.h, tb = (size_t)tmpl.w * tmpl.h, rb = (size_t)result.w * result.h;
  const uint8_t *s = (const uint8_t *)stage_in(img.data, ib, SL_IN);
  const uint8_t *t = (const uint8_t *)stage_in(tmpl.data, tb, SL_AUX);
  const bool dhost = !is_dev(result.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, rb) : result.data;
  const dim3 g((result.w + 63) / 64, (result.h + 3) / 4);
  /* templates of 512 .. 32768 taps, 16 .. 257 wide, whose block fits the LDS: the cross term on the matrix cores (k_tmatch.h;
   * smaller ones are as fast on the dot-product kernels: 1280x720, 16 x 16: 39 vs 47 us).
   * gsh_tune key 20 = 1: the VALU dot-product kernels below; 2 / 3: always 64 x 128 tiles / always 32 x 64 tiles with split rows. */
  /* few 64 x 128 tiles (video-sized images): 32 x 64 tiles, the template rows split over the block's four waves */
  const bool tm_split = g_tune[20] == 3 || (g_tune[20] != 2 && g_tune[20] != 8 && (unsigned long long)((result.w + 127) / 128) * ((result.h + 63) / 64) < 512); /* key 20 = 8: 64 x 128 tiles, banded, whatever the size */
  const unsigned nkc = (tmpl.w + 31 + 31) / 32, istride = (tm_split ? 32 : 96) + 32 * nkc + 16, tstride = 32 * nkc + 48;
  /* SPLIT = 1 takes templates taller than a band kTmBand rows at a time (k_tmatch.h BAND: 33 KB of LDS per block instead of 79 at
   * 128 x 128, four blocks per CU); key 20 = 2: the whole template resident, as in round 3 */
  const bool tm_band = !tm_split && g_tune[20] != 2 && tmpl.h > kTmBand;
  const unsigned tm_rows = tm_band ? kTmBand : tmpl.h;
  const size_t tm_lds = std::max<size_t>((size_t)((tm_split ? 31 : 63) + tm_rows) * istride + (size_t)tm_rows * tstride + 16,
                                         tm_split ? 32768 : 0);
  if (g_tune[20] != 1 && tmpl.w >= 16 && nkc <= 9 && tmpl.h >= 4 && (tb >= 512 || g_tune[20] == 2 || g_tune[20] == 3 || g_tune[20] == 8) && tb <= 32768 && tm_lds <= 150 * 1024 &&
      ib < 0x7fffffffull) {
    hipStream_t st = ctx().s();
    unsigned *rowp = (unsigned *)ctx().scratch(SL_II, (size_t)img.h * (img.w + 1) * 4);
    unsigned *s2 = (unsigned *)ctx().scratch(SL_PAD, rb * 4);
    uint8_t *tpad = (uint8_t *)ctx().scratch(SL_PRE, (size_t)tmpl.h * tstride + 16);
    unsigned *tsqp = (unsigned *)(tpad + (((size_t)tmpl.h * tstride + 3) & ~(size_t)3)); /* tstride is a multiple of 16 */
    GS_LAUNCH(k_tm_prep, dim3(1), dim3(1024), 0, st, t, tmpl.w, tmpl.h, tstride, tpad, tsqp);
    /* window sums of (I - 128)^2: four corners of the banded integral table of squares (round 4) for frames of 4 Mpx and
     * more -- 4K, 128 x 128: 0.269 -> 0.249 ms, 64 x 64: 0.135 -> 0.125; on a 720p frame the four short launches are 6-10 us
     * SLOWER than the two passes of round 3 (row prefix + sliding columns), which stay for small frames, for key 20 = 4 and
     * for geometries the banded integral does not take (key 20 = 5: the table route whatever the size);
     * profiles/r04v_match_template_integral_squares.log */
    if (g_tune[20] != 4 && (ib >= (4u << 20) || g_tune[20] == 5) && launch_integral(s, img.w, img.h, 1, rowp, /*sq=*/true)) {
      GS_LAUNCH(k_tm_s2_corners, dim3((result.w + 255) / 256, result.h), dim3(256), 0, st, (const unsigned *)rowp, img.w, tmpl.w, tmpl.h,
                result.w, result.h, s2);
    } else {
      GS_LAUNCH(k_tm_rowprefix, dim3((img.h + 3) / 4), dim3(64, 4), 0, st, s, img.w, img.h, rowp);
      GS_LAUNCH(k_tm_colsq, dim3((result.w + 63) / 64, (result.h + kTmRun - 1) / kTmRun), dim3(64), 0, st, (const unsigned *)rowp,
                img.w, tmpl.w, tmpl.h, result.w, result.h, s2);
    }
#ifndef GS_EMU
    /* more than the default 64 KB of dynamic LDS: a per-DEVICE attribute of the function (a thread that moved to
     * another device with gsh_set_device kept a thread-local "done" flag and the launch failed there) */
    static std::atomic<unsigned long long> lds_raised{0};
    const unsigned long long dev_bit = 1ull << ((unsigned)ctx().device & 63u);
    if (ctx().device >= 64 || !(lds_raised.load(std::memory_order_acquire) & dev_bit)) {
#define GS_TM_FNS(S) (const void *)k_match_template_mfma<S, 2>, (const void *)k_match_template_mfma<S, 3>, (const void *)k_match_template_mfma<S, 4>, \
                     (const void *)k_match_template_mfma<S, 5>, (const void *)k_match_template_mfma<S, 6>, (const void *)k_match_template_mfma<S, 7>, \
                     (const void *)k_match_template_mfma<S, 8>, (const void *)k_match_template_mfma<S, 9>
      const void *fns[] = {GS_TM_FNS(1), GS_TM_FNS(4),
                           (const void *)k_match_template_mfma<1, 2, true>, (const void *)k_match_template_mfma<1, 3, true>, (const void *)k_match_template_mfma<1, 4, true>,
                           (const void *)k_match_template_mfma<1, 5, true>, (const void *)k_match_template_mfma<1, 6, true>, (const void *)k_match_template_mfma<1, 7, true>,
                           (const void *)k_match_template_mfma<1, 8, true>, (const void *)k_match_template_mfma<1, 9, true>};
#undef GS_TM_FNS
      for (const void *fn : fns) GS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      lds_raised.fetch_or(dev_bit, std::memory_order_release);
    }
#endif
    TmArgs ta{s, img.w, img.h, (const uint8_t *)tpad, (const unsigned *)tsqp, tmpl.w, tmpl.h, s2, d, result.w, result.h, nkc, istride, tstride};
    /* one instantiation per number of K steps (k_tmatch.h): nkc = 2 .. 9 for templates 16 .. 257 wide */
    const dim3 gs4((result.w + 63) / 64, (result.h + 31) / 32), gs1((result.w + 127) / 128, (result.h + 63) / 64);
    GS_ASSERT(nkc >= 2 && nkc <= 9);
    switch (nkc * 4 + (tm_split ? 1u : tm_band ? 2u : 0u)) {
#define GS_TM_CASE(K)                                                                                   \
  case K * 4: GS_LAUNCH((k_match_template_mfma<1, K>), gs1, dim3(256), tm_lds, st, ta); break;          \
  case K * 4 + 1: GS_LAUNCH((k_match_template_mfma<4, K>), gs4, dim3(256), tm_lds, st, ta); break;      \
  case K * 4 + 2: GS_LAUNCH((k_match_template_mfma<1, K, true>), gs1, dim3(256), tm_lds, st, ta); break;
      GS_TM_CASE(2) GS_TM_CASE(3) GS_TM_CASE(4) GS_TM_CASE(5) GS_TM_CASE(6) GS_TM_CASE(7) GS_TM_CASE(8) GS_TM_CASE(9)
#undef GS_TM_CASE
    }
  } else if (tmpl.w <= kTmplTile - 3) {
    unsigned long long *tsq = (unsigned long long *)ctx().scratch(SL_PFX, 8);
    GS_LAUNCH(k_sum_squares, dim3(1), dim3(256), 0, ctx().s(), t, (unsigned long long)tb, tsq);
    const size_t twp = ((size_t)tmpl.w + 3) & ~(size_t)3;
    const size_t lds = std::min<size_t>(twp * tmpl.h, (kTmplTile / twp) * twp);
    /* four results per thread when image rows start 4-byte aligned (see k_match_template4) */
    if (img.w % 4 == 0 && ((uintptr_t)s & 3) == 0 && (size_t)img.w * img.h < 0x7fffffffull)
      GS_LAUNCH(k_match_template4, dim3((result.w + 255) / 256, (result.h + 3) / 4), dim3(64, 4), lds, ctx().s(),
                s, img.w, img.h, t, tmpl.w, tmpl.h, (const unsigned long long *)tsq, d, result.w, result.h);
    else
      GS_LAUNCH(k_match_template, g, dim3(64, 4), lds, ctx().s(), s, img.w, img.h, t, tmpl.w, tmpl.h,
                (const unsigned long long *)tsq, d, result.w, result.h);
  } else {
    GS_LAUNCH(k_match_template_px, g, dim3(64, 4), 0, ctx().s(), s, img.w, img.h, t, tmpl.w, tmpl.h, d,
              result.w, result.h);
  }
  if (dhost) GS_HIP(hipMemcpyAsync(result.data, d, rb, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}
struct gs_point gs_find_best_match(struct gs_image result) { /* ref :726 */
  GS_ASSERT(GS_VALID(result));
  const unsigned long long n = (unsigned long long)result.w * result.h;
  const uint8_t *s = (const uint8_t *)stage_in(result.data, (size_t)n, SL_IN);
  const unsigned blocks = (unsigned)((n + 2047) / 2048);
  unsigned long long *part = (unsigned long long *)ctx().scratch(SL_PFX, (size_t)blocks * 8);
  GS_LAUNCH(k_argmax_first, dim3(blocks), dim3(256), 0, ctx().s(), s, n, part);
  std::vector<unsigned long long> hp(blocks);
  GS_HIP(hipMemcpyAsync(hp.data(), part, (size_t)blocks * 8, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  unsigned long long best = 0;
  for (unsigned long long k : hp) best = std::max(best, k);
  struct gs_point p = {0, 0};
  if (best >> 32) { /* a zero maximum leaves the reference's initial {0,0} */
    const unsigned idx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
    p.x = idx % result.w, p.y = idx / result.w;
  }
  return p;
}

void gs_histogram(struct gs_image img, unsigned hist[256]) { /* ref :199 */
  GS_ASSERT(GS_VALID(img) && hist != NULL);
  const size_t nb = (size_t)(img.w * img.h);
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool hhost = !is_dev(hist);
  unsigned *dh = hhost ? (unsigned *)ctx().scratch(SL_HIST, 1024) : hist;
  launch_histogram(s, nb, 1, dh);
  if (hhost) GS_HIP(hipMemcpyAsync(hist, dh, 1024, hipMemcpyDeviceToHost, ctx().s()));
  finish(hhost);
}

uint8_t gs_otsu_threshold(struct gs_image img) { /* ref :205 */
  GS_ASSERT(GS_VALID(img));
  const size_t nb = (size_t)(img.w * img.h);
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  unsigned *dh = (unsigned *)ctx().scratch(SL_HIST, 1024);
  uint8_t *dt = (uint8_t *)ctx().scratch(SL_THR, 16);
  launch_otsu(s, img.w, img.h, 1, dh, dt);
  uint8_t t = 0;
  GS_HIP(hipMemcpyAsync(&t, dt, 1, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  return t;
}

void gs_threshold(struct gs_image img, uint8_t thresh) { /* ref :225 */
  GS_ASSERT(GS_VALID(img));
  const size_t nb = (size_t)(img.w * img.h);
  const bool host = !is_dev(img.data);
  uint8_t *d = host ? (uint8_t *)ctx().scratch(SL_IN, nb) : img.data;
  if (host) GS_HIP(hipMemcpyAsync(d, img.data, nb, hipMemcpyHostToDevice, ctx().s()));
  launch_threshold(d, nb, 1, nullptr, thresh);
  if (host) GS_HIP(hipMemcpyAsync(img.data, d, nb, hipMemcpyDeviceToHost, ctx().s()));
  finish(host);
}

void gs_integral(struct gs_image src, unsigned *ii) { /* ref :744 */
  GS_ASSERT(GS_VALID(src) && ii);
  const size_t np = (size_t)src.w * src.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, np, SL_IN);
  const bool host = !is_dev(ii);
  unsigned *d = host ? (unsigned *)ctx().scratch(SL_II, np * 4) : ii;
  launch_integral(s, src.w, src.h, 1, d);
  if (host) GS_HIP(hipMemcpyAsync(ii, d, np * 4, hipMemcpyDeviceToHost, ctx().s()));
  finish(host);
}

}  /* extern "C" */
