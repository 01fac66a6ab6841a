/*
 * include/grayskull.h -- drop-in replacement for the HOT PATH of zserge/grayskull's
 * single header, backed by hand-written HIP kernels for AMD MI355X (gfx950).
 *
 * The reference declares every public function `GS_API ret name(...)` with
 * `GS_API` defaulting to `static inline` (reference grayskull.h:7-9); the wasm
 * example already overrides that seam to get external linkage
 * (examples/wasm/grayskull.c:31-34).  This header uses the same seam: the structs
 * are byte-identical (reference grayskull.h:14-64), the hot-path functions become
 * `extern` symbols of libgrayskull_hip.so, and the trivial per-pixel helpers
 * (gs_valid / gs_get / gs_set / gs_for / gs_integral_sum) stay header inlines.
 *
 * Host code stays C99:   cc -std=c99 -Iinclude app.c -Lgrayskull_amd -lgrayskull_hip
 *
 * `gs_image.data` (and every other buffer argument) may be
 *   - a HOST pointer: the call stages it through device scratch and is
 *     synchronous, exactly like the reference call it replaces; or
 *   - a DEVICE pointer (hipMalloc / gsh_malloc): zero-copy, stream-ordered on the
 *     calling thread's stream (see grayskull_hip.h), synchronised before return
 *     unless gsh_set_async(1).
 *
 * Out of scope here (not on the hot path): gs_blobs, contours,
 * perspective correction, PGM I/O, gs_alloc / gs_free.  Callers that need them
 * include the reference header for those functions.
 */
#ifndef GRAYSKULL_H
#define GRAYSKULL_H

#include <limits.h>
#include <stdint.h>
#ifndef GS_NO_STDLIB /* ref :68: "no asserts, no memory allocation, no file I/O" */
#include <stdio.h>
#include <stdlib.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#ifndef GS_API
#define GS_API extern
#endif

#define GS_MIN(a, b) ((a) < (b) ? (a) : (b))
#define GS_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ---- types: identical layout to reference grayskull.h:14-64 -------------------- */
struct gs_image { unsigned w, h; uint8_t *data; };                 /* 16 B, ref :14-17 */
struct gs_rect { unsigned x, y, w, h; };                           /* 16 B, ref :19-21 */
struct gs_point { unsigned x, y; };                                /*  8 B, ref :23-25 */
struct gs_keypoint {                                               /* 48 B, ref :42-47 */
  struct gs_point pt;
  unsigned response;
  float angle;
  uint32_t descriptor[8];
};
struct gs_match { unsigned idx1, idx2; unsigned distance; };       /* 12 B, ref :49-52 */
struct gs_lbp_cascade {                                            /* 96 B, ref :54-64 */
  uint16_t window_w, window_h;
  uint16_t nfeatures, nweaks, nstages;
  const int8_t *features; /* [nfeatures * 4] */
  const uint16_t *weak_feature_idx;
  const float *weak_left_val, *weak_right_val;
  const uint16_t *weak_subset_offset, *weak_num_subsets;
  const int32_t *subsets;
  const uint16_t *stage_weak_start, *stage_nweaks;
  const float *stage_threshold;
};

/* ---- header inlines (host side; ref :66, :139-148, :754-763) -------------------- */
static inline int gs_valid(struct gs_image img) { return img.data && img.w > 0 && img.h > 0; }

#define gs_for(img, x, y)                \
  for (unsigned y = 0; y < (img).h; y++) \
    for (unsigned x = 0; x < (img).w; x++)

/* gs_get/gs_set touch img.data directly: only meaningful for HOST images. */
static inline uint8_t gs_get(struct gs_image img, unsigned x, unsigned y) {
  return (gs_valid(img) && x < img.w && y < img.h) ? img.data[y * img.w + x] : 0;
}
static inline void gs_set(struct gs_image img, unsigned x, unsigned y, uint8_t value) {
  if (gs_valid(img) && x < img.w && y < img.h) img.data[y * img.w + x] = value;
}
static inline uint32_t gs_integral_sum(const unsigned *ii, unsigned iw, unsigned x, unsigned y,
                                       unsigned w, unsigned h) {
#ifndef GS_NO_STDLIB /* gs_assert of ref :756, same message and abort(); compiled out under GS_NO_STDLIB like ref :69 */
  if (!(ii && iw > 0 && x + w <= iw)) {
    fprintf(stderr, "Assertion failed: %s\n", "ii && iw > 0 && x + w <= iw");
    abort();
  }
#endif
  unsigned x2 = x + w - 1, y2 = y + h - 1;
  unsigned A = (x > 0 && y > 0) ? ii[(y - 1) * iw + (x - 1)] : 0;
  unsigned B = (y > 0) ? ii[(y - 1) * iw + x2] : 0;
  unsigned C = (x > 0) ? ii[y2 * iw + (x - 1)] : 0;
  return ii[y2 * iw + x2] + A - B - C;
}

/* ---- hot path: same names, parameter order and error behaviour as the reference.
 *      Precondition failures print "Assertion failed: <cond>" and abort(), like
 *      gs_assert (ref :94-98).  Each prototype cites the definition it replaces. --- */

/* stencils */
GS_API void gs_blur(struct gs_image dst, struct gs_image src, unsigned radius); /* ref :268 */
GS_API void gs_sobel(struct gs_image dst, struct gs_image src);                 /* ref :306 */
GS_API void gs_erode(struct gs_image dst, struct gs_image src);                 /* ref :303 */
GS_API void gs_dilate(struct gs_image dst, struct gs_image src);                /* ref :304 */

/* histogram / threshold */
GS_API void gs_histogram(struct gs_image img, unsigned hist[256]);              /* ref :199 */
GS_API uint8_t gs_otsu_threshold(struct gs_image img);                          /* ref :205 */
GS_API void gs_threshold(struct gs_image img, uint8_t thresh);                  /* ref :225 */

/* integral image + LBP cascade */
GS_API void gs_integral(struct gs_image src, unsigned *ii);                     /* ref :744 */
GS_API unsigned gs_lbp_window(const struct gs_lbp_cascade *c, const unsigned *ii, unsigned iw,
                              unsigned ih, int x, int y, float scale);          /* ref :790 */
GS_API unsigned gs_lbp_detect(const struct gs_lbp_cascade *c, const unsigned *ii, unsigned iw,
                              unsigned ih, struct gs_rect *rects, unsigned max_rects,
                              float scale_factor, float min_scale, float max_scale,
                              int step);                                        /* ref :815 */

/* FAST / ORB / matching */
GS_API unsigned gs_fast(struct gs_image img, struct gs_image scoremap, struct gs_keypoint *kps,
                        unsigned nkps, unsigned threshold);                     /* ref :482 */
GS_API float gs_compute_orientation(struct gs_image img, unsigned x, unsigned y,
                                    unsigned r);                                /* ref :608 */
GS_API void gs_brief_descriptor(struct gs_image img, struct gs_keypoint *kp);   /* ref :623 */
GS_API unsigned gs_orb_extract(struct gs_image img, struct gs_keypoint *kps, unsigned nkps,
                               unsigned threshold, uint8_t *scoremap_buffer);   /* ref :651 */
GS_API unsigned gs_match_orb(const struct gs_keypoint *kps1, unsigned n1,
                             const struct gs_keypoint *kps2, unsigned n2,
                             struct gs_match *matches, unsigned max_matches,
                             float max_distance);                               /* ref :680 */

/* The reference has TWO trig flavours, chosen at compile time (ref :68-101): libm's atan2f / sinf, or -- under
 * GS_NO_STDLIB, what examples/wasm/grayskull.c:31-35 builds -- two float32 polynomials (ref :70-88) with gs_assert
 * compiled out (ref :69).  Angles and, through the (int) truncation of ref :633, descriptor bits differ between the
 * two.  A caller built -DGS_NO_STDLIB therefore gets the polynomial flavour here as well, so that it stays bit-exact
 * against ITS OWN reference build: the three functions that reach gs_atan2 / gs_sin bind to the *_nostdlib symbols
 * (polynomials evaluated on the device for gs_orb_extract, no host round trip between FAST, selection, orientation and
 * BRIEF; no precondition aborts).  Everything else is integer arithmetic and identical in both flavours. */
GS_API float gs_compute_orientation_nostdlib(struct gs_image img, unsigned x, unsigned y, unsigned r);
GS_API void gs_brief_descriptor_nostdlib(struct gs_image img, struct gs_keypoint *kp);
GS_API unsigned gs_orb_extract_nostdlib(struct gs_image img, struct gs_keypoint *kps, unsigned nkps,
                                        unsigned threshold, uint8_t *scoremap_buffer);
#ifdef GS_NO_STDLIB
#define gs_compute_orientation gs_compute_orientation_nostdlib
#define gs_brief_descriptor gs_brief_descriptor_nostdlib
#define gs_orb_extract gs_orb_extract_nostdlib
#endif

/* "next" rows of SURVEY.md 8(f), same machinery as the stencils above */
GS_API void gs_adaptive_threshold(struct gs_image dst, struct gs_image src, unsigned radius,
                                  int c);                                       /* ref :230 */
GS_API void gs_filter(struct gs_image dst, struct gs_image src, struct gs_image kernel,
                      unsigned norm);                                           /* ref :255 */
GS_API void gs_downsample(struct gs_image dst, struct gs_image src);            /* ref :189 */

/* ---- geometry helpers and template matching ----------------------- */
GS_API void gs_crop(struct gs_image dst, struct gs_image src, struct gs_rect roi); /* ref :154 */
GS_API void gs_copy(struct gs_image dst, struct gs_image src);                  /* ref :160 */
GS_API void gs_resize_nn(struct gs_image dst, struct gs_image src);             /* ref :164 */
GS_API void gs_resize(struct gs_image dst, struct gs_image src);                /* ref :171, float32 bilinear */
GS_API void gs_match_template(struct gs_image img, struct gs_image tmpl,
                              struct gs_image result);                          /* ref :705 */
GS_API struct gs_point gs_find_best_match(struct gs_image result);              /* ref :726 */

/* 3x3 kernels for gs_filter, as the reference spells them (ref :249-253) */
#define gs_sharpen ((struct gs_image){3, 3, (uint8_t[]){0, -1, 0, -1, 5, -1, 0, -1, 0}})
#define gs_emboss ((struct gs_image){3, 3, (uint8_t[]){-2, -1, 0, -1, 1, 1, 0, 1, 2}})
#define gs_blur_box ((struct gs_image){3, 3, (uint8_t[]){1, 1, 1, 1, 1, 1, 1, 1, 1}})
#define gs_blur_gaussian ((struct gs_image){3, 3, (uint8_t[]){1, 2, 1, 2, 4, 2, 1, 2, 1}})

#ifdef __cplusplus
}
#endif
#endif /* GRAYSKULL_H */
