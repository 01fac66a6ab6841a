/*
 * tests/c/test_dropin.c -- a C99 host program using the drop-in header exactly the way the
 * reference's own unit tests use theirs (reference test.c: test_blur :72, test_morph :88,
 * test_sobel :121, test_histogram :151, test_threshold :167, test_otsu :177,
 * test_adaptive_threshold :198, test_integral :289).  Vectors are restated, not copied.
 * Build: cc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude tests/c/test_dropin.c -L... -lgrayskull_hip
 */
#include <assert.h>
#include <stdio.h>
#include <string.h>

#include "grayskull.h"

#define W 255

static void t_blur(void) {
  uint8_t s[9] = {0, 0, 0, 0, W, 0, 0, 0, 0}, d[9];
  struct gs_image src = {3, 3, s}, dst = {3, 3, d};
  gs_blur(dst, src, 1);
  assert(d[4] == 28); /* 255/9 */
  assert(d[0] == 63); /* 255/4: corner averages the 4 in-image taps */
}

static void t_morph(void) {
  uint8_t e[25] = {0}, d[25];
  for (int y = 1; y < 4; y++)
    for (int x = 1; x < 4; x++) e[y * 5 + x] = W;
  struct gs_image src = {5, 5, e}, dst = {5, 5, d};
  gs_erode(dst, src);
  assert(d[12] == 255 && d[6] == 0);
  uint8_t p[25] = {0};
  p[12] = W;
  src.data = p;
  gs_dilate(dst, src);
  assert(d[12] == 255 && d[7] == 255 && d[17] == 255 && d[11] == 255 && d[13] == 255 && d[0] == 0);
}

static void t_sobel(void) {
  uint8_t s[25], d[25];
  for (int y = 0; y < 5; y++)
    for (int x = 0; x < 5; x++) s[y * 5 + x] = x >= 2 ? W : 0;
  memset(d, 0xAB, sizeof d);
  struct gs_image src = {5, 5, s}, dst = {5, 5, d};
  gs_sobel(dst, src);
  assert(d[12] > 100 && d[11] > 100 && d[13] == 0);
  for (int y = 0; y < 5; y++)
    for (int x = 0; x < 5; x++)
      if (x == 0 || y == 0 || x == 4 || y == 4) assert(d[y * 5 + x] == 0xAB); /* frame untouched */
}

static void t_hist_thresh_otsu(void) {
  uint8_t s[6] = {0, 0, 128, 128, 255, 255};
  struct gs_image img = {3, 2, s};
  unsigned hist[256];
  gs_histogram(img, hist);
  assert(hist[0] == 2 && hist[128] == 2 && hist[255] == 2 && hist[1] == 0);
  uint8_t t[4] = {100, 150, 200, 50};
  struct gs_image ti = {4, 1, t};
  gs_threshold(ti, 128);
  assert(t[0] == 0 && t[1] == 255 && t[2] == 255 && t[3] == 0);
  uint8_t b[16];
  for (int i = 0; i < 16; i++) b[i] = i < 8 ? 50 : 200;
  struct gs_image bi = {4, 4, b};
  uint8_t o = gs_otsu_threshold(bi);
  assert(o >= 50 && o < 200);
  memset(b, 77, sizeof b);
  assert(gs_otsu_threshold(bi) == 0); /* constant image */
}

static void t_adaptive(void) {
  uint8_t s[25], d[25];
  memset(s, 100, sizeof s);
  struct gs_image src = {5, 5, s}, dst = {5, 5, d};
  gs_adaptive_threshold(dst, src, 1, 5);
  for (int i = 0; i < 25; i++) assert(d[i] == 255);
  gs_adaptive_threshold(dst, src, 1, -5);
  for (int i = 0; i < 25; i++) assert(d[i] == 0);
}

static void t_integral(void) {
  uint8_t s[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
  unsigned ii[9], want[9] = {1, 3, 6, 5, 12, 21, 12, 27, 45};
  struct gs_image src = {3, 3, s};
  gs_integral(src, ii);
  for (int i = 0; i < 9; i++) assert(ii[i] == want[i]);
  assert(gs_integral_sum(ii, 3, 1, 1, 2, 2) == 28);
}

/* SURVEY 8(f) rank 4 through the same header: crop / resize / template matching */
static void t_geometry(void) {
  uint8_t s[16], c[4], n[16], b[4], line[2] = {0, W};
  for (int i = 0; i < 16; i++) s[i] = (uint8_t)(10 * i);
  struct gs_image src = {4, 4, s}, crop = {2, 2, c}, nn = {4, 4, n};
  struct gs_rect roi = {1, 2, 2, 2};
  gs_crop(crop, src, roi); /* rows 2..3, columns 1..2 */
  assert(c[0] == 90 && c[1] == 100 && c[2] == 130 && c[3] == 140);
  uint8_t two[4] = {1, 2, 3, 4};
  struct gs_image small = {2, 2, two};
  gs_resize_nn(nn, small); /* every source pixel becomes a 2x2 block */
  assert(n[0] == 1 && n[1] == 1 && n[2] == 2 && n[5] == 1 && n[10] == 4 && n[15] == 4);
  /* bilinear 2x1 -> 4x1 with centres at +0.5: sample positions -0.25 (clamped), 0.25, 0.75, 1.25 (clamped) */
  struct gs_image l2 = {2, 1, line}, l4 = {4, 1, b};
  gs_resize(l4, l2);
  assert(b[0] == 0 && b[1] == 63 && b[2] == 191 && b[3] == 255);
}

static void t_template(void) {
  uint8_t img[64] = {0}, tm[4] = {W, 40, 40, W}, res[49];
  img[3 * 8 + 5] = W, img[3 * 8 + 6] = 40, img[4 * 8 + 5] = 40, img[4 * 8 + 6] = W; /* the patch at (5,3) */
  struct gs_image im = {8, 8, img}, t = {2, 2, tm}, r = {7, 7, res};
  gs_match_template(im, t, r);
  assert(res[3 * 7 + 5] == 255);        /* zero squared difference */
  assert(res[0] < 255);                 /* an all-zero window differs */
  struct gs_point p = gs_find_best_match(r);
  assert(p.x == 5 && p.y == 3);
}

int main(void) {
  t_blur();
  t_morph();
  t_sobel();
  t_hist_thresh_otsu();
  t_adaptive();
  t_integral();
  t_geometry();
  t_template();
  printf("dropin C99 tests: all passed\n");
  return 0;
}
