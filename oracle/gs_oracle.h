/*
 * oracle/gs_oracle.h -- CPU oracle for the Grayskull pixel-array hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under grayskull_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker.
 *
 * Every function is a plain-C restatement of the reference algorithm with flat
 * (pointer, size) arguments; the reference location it follows is cited at each
 * definition in gs_oracle.c (paths relative to the reference checkout).
 *
 * Parity pinning: tests/test_oracle.py checks this file against
 *   (1) the reference's own unit-test vectors (test.c:72-196, 289-307),
 *   (2) the known-answer hashes of SURVEY.md 8(c) (tests/golden/kat.json),
 *   (3) the compiled reference itself (oracle/_ref/libgs_ref.so) on random inputs.
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H
#include <stdint.h>

typedef struct { uint32_t x, y, response; float angle; uint32_t desc[8]; } orc_keypoint; /* 48 B */
typedef struct { uint32_t x, y, w, h; } orc_rect;                                        /* 16 B */
typedef struct { uint32_t idx1, idx2, distance; } orc_match;                             /* 12 B */

/* same field order/types as struct gs_lbp_cascade (grayskull.h:54-64) */
typedef struct {
  uint16_t window_w, window_h, nfeatures, nweaks, nstages;
  const int8_t *features;
  const uint16_t *weak_feature_idx;
  const float *weak_left_val, *weak_right_val;
  const uint16_t *weak_subset_offset, *weak_num_subsets;
  const int32_t *subsets;
  const uint16_t *stage_weak_start, *stage_nweaks;
  const float *stage_threshold;
} orc_cascade;

/* synthetic frames + hashing */
void orc_synth(uint8_t *img, unsigned w, unsigned h, uint32_t seed);
uint32_t orc_fnv1a(const void *data, uint64_t nbytes);

/* stencils / pointwise */
void orc_blur(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned radius);
void orc_sobel(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h);
void orc_erode(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h);
void orc_dilate(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h);
void orc_histogram(const uint8_t *img, unsigned w, unsigned h, unsigned hist[256]);
uint8_t orc_otsu_from_hist(const unsigned hist[256], unsigned npix);
uint8_t orc_otsu_threshold(const uint8_t *img, unsigned w, unsigned h);
void orc_threshold(uint8_t *img, unsigned w, unsigned h, uint8_t t);
void orc_adaptive_threshold(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                            unsigned radius, int c);
void orc_filter(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, const uint8_t *kernel,
                unsigned kw, unsigned kh, unsigned norm);
void orc_downsample(uint8_t *dst, const uint8_t *src, unsigned sw, unsigned sh);

/* geometry + template matching */
void orc_crop(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src, unsigned sw, unsigned sh,
              unsigned rx, unsigned ry, unsigned rw, unsigned rh);
void orc_resize_nn(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src, unsigned sw, unsigned sh);
void orc_resize(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src, unsigned sw, unsigned sh);
void orc_match_template(const uint8_t *img, unsigned iw, unsigned ih, const uint8_t *tmpl, unsigned tw,
                        unsigned th, uint8_t *result);
void orc_find_best_match(const uint8_t *result, unsigned w, unsigned h, unsigned *bx, unsigned *by);

/* integral image + LBP cascade */
void orc_integral(const uint8_t *src, unsigned w, unsigned h, unsigned *ii);
unsigned orc_integral_sum(const unsigned *ii, unsigned iw, unsigned x, unsigned y, unsigned w,
                          unsigned h);
unsigned orc_lbp_window(const orc_cascade *c, const unsigned *ii, unsigned iw, unsigned ih, int x,
                        int y, float scale);
unsigned orc_lbp_detect(const orc_cascade *c, const unsigned *ii, unsigned iw, unsigned ih,
                        orc_rect *rects, unsigned max_rects, float scale_factor, float min_scale,
                        float max_scale, int step);

/* FAST / ORB / matching */
unsigned orc_fast(const uint8_t *img, unsigned w, unsigned h, uint8_t *scoremap, orc_keypoint *kps,
                  unsigned nkps, unsigned threshold);
/* 1: the GS_NO_STDLIB polynomials of ref :70-88 for gs_atan2 / gs_sin; 0 (default): libm, ref :100-101 */
void orc_set_nostdlib(int on);
float orc_atan2_poly(float y, float x);
float orc_sin_poly(float x);
float orc_orientation(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y,
                      unsigned r);
void orc_orientation_moments(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y,
                             unsigned r, float *m01, float *m10);
void orc_brief(const uint8_t *img, unsigned w, unsigned h, orc_keypoint *kp);
unsigned orc_orb_extract(const uint8_t *img, unsigned w, unsigned h, orc_keypoint *kps,
                         unsigned nkps, unsigned threshold, uint8_t *scoremap);
/* ref examples/nanomagick/nanomagick.c:245-290; buffer: see orc_orb_pyramid_buffer_bytes */
unsigned orc_orb_extract_pyramid(const uint8_t *img, unsigned w, unsigned h, orc_keypoint *kps,
                                 unsigned nkps, unsigned threshold, uint8_t *buffer, unsigned n_levels);
unsigned orc_match_orb(const orc_keypoint *k1, unsigned n1, const orc_keypoint *k2, unsigned n2,
                       orc_match *out, unsigned max_matches, float max_distance);
#endif
