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

int main(void) {
  t_blur();
  t_morph();
  t_sobel();
  t_hist_thresh_otsu();
  t_adaptive();
  t_integral();
  printf("dropin C99 tests: all passed\n");
  return 0;
}
