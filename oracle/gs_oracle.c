/*
 * oracle/gs_oracle.c -- CPU oracle (plain C99 restatement of the reference algorithms).
 *
 * TEST INFRASTRUCTURE ONLY -- see gs_oracle.h.  Build: `make -C oracle` with
 * -std=c99 -O2 (ISO mode => no FMA contraction, like the reference Makefile:1).
 *
 * All "ref:" citations are file:line in the reference checkout (grayskull.h unless
 * another file is named).  Pixel access everywhere follows gs_get/gs_set
 * (ref :143-148): out-of-range reads give 0, out-of-range writes are dropped.
 */
#include "gs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int8_t k_brief[1024] = {
#include "brief_pattern.inc"
};

/* ref :143-145 -- unsigned compare makes "negative" coordinates out of range */
static inline uint8_t px(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y) {
  return (img && w && h && x < w && y < h) ? img[(size_t)y * w + x] : 0;
}

/* ---------------------------------------------------------------- synthetic frames */
/* xorshift32, one level per 32x32 block, noise in [-8,7], clamp. */
static uint32_t xorshift32(uint32_t *s) {
  uint32_t v = *s;
  v ^= v << 13;
  v ^= v >> 17;
  v ^= v << 5;
  return *s = v;
}

void orc_synth(uint8_t *img, unsigned w, unsigned h, uint32_t seed) {
  uint32_t s = seed ? seed : 1u;
  unsigned bw = (w + 31) / 32, bh = (h + 31) / 32;
  uint8_t *level = (uint8_t *)malloc((size_t)bw * bh);
  for (unsigned i = 0; i < bw * bh; i++) level[i] = (uint8_t)(xorshift32(&s) & 0xFF);
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++) {
      int v = (int)level[(y / 32) * bw + x / 32] + (int)(xorshift32(&s) & 15u) - 8;
      img[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(level);
}

uint32_t orc_fnv1a(const void *data, uint64_t nbytes) {
  const uint8_t *p = (const uint8_t *)data;
  uint32_t hsh = 2166136261u;
  for (uint64_t i = 0; i < nbytes; i++) hsh = (hsh ^ p[i]) * 16777619u;
  return hsh;
}

/* ---------------------------------------------------------------- box blur */
/* ref :268-283 -- mean over the (2r+1)^2 window clipped to the image, unsigned
 * truncating division by the number of in-image taps; every pixel written. */
void orc_blur(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned radius) {
  long r = (long)radius;
  for (long y = 0; y < (long)h; y++) {
    long ya = y - r < 0 ? 0 : y - r, yb = y + r > (long)h - 1 ? (long)h - 1 : y + r;
    for (long x = 0; x < (long)w; x++) {
      long xa = x - r < 0 ? 0 : x - r, xb = x + r > (long)w - 1 ? (long)w - 1 : x + r;
      unsigned sum = 0; /* wraps mod 2^32 like the reference's `unsigned sum` */
      for (long yy = ya; yy <= yb; yy++)
        for (long xx = xa; xx <= xb; xx++) sum += src[(size_t)yy * w + xx];
      unsigned count = (unsigned)((xb - xa + 1) * (yb - ya + 1));
      dst[(size_t)y * w + x] = (uint8_t)(sum / count);
    }
  }
}

/* ---------------------------------------------------------------- sobel */
/* ref :306-320 -- interior only (1-px frame of dst untouched), (|gx|+|gy|)/2 clamped */
void orc_sobel(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h) {
  if (w < 3 || h < 3) return;
  for (unsigned y = 1; y + 1 < h; y++) {
    const uint8_t *a = src + (size_t)(y - 1) * w, *b = a + w, *c = b + w;
    for (unsigned x = 1; x + 1 < w; x++) {
      int gx = (a[x + 1] - a[x - 1]) + 2 * (b[x + 1] - b[x - 1]) + (c[x + 1] - c[x - 1]);
      int gy = (c[x - 1] + 2 * c[x] + c[x + 1]) - (a[x - 1] + 2 * a[x] + a[x + 1]);
      int mag = (abs(gx) + abs(gy)) / 2;
      dst[(size_t)y * w + x] = (uint8_t)(mag > 255 ? 255 : mag);
    }
  }
}

/* ---------------------------------------------------------------- morphology */
/* ref :285-304 -- 3x3 min (erode, init 255) / max (dilate, init 0) over in-image taps */
static void morph3(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, int dilate) {
  for (long y = 0; y < (long)h; y++)
    for (long x = 0; x < (long)w; x++) {
      uint8_t v = dilate ? 0 : 255;
      for (long yy = y - 1; yy <= y + 1; yy++)
        for (long xx = x - 1; xx <= x + 1; xx++) {
          if (yy < 0 || yy >= (long)h || xx < 0 || xx >= (long)w) continue;
          uint8_t p = src[(size_t)yy * w + xx];
          if (dilate ? p > v : p < v) v = p;
        }
      dst[(size_t)y * w + x] = v;
    }
}
void orc_erode(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h) { morph3(dst, src, w, h, 0); }
void orc_dilate(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h) { morph3(dst, src, w, h, 1); }

/* ---------------------------------------------------------------- histogram / otsu / threshold */
/* ref :199-203 -- note the 32-bit product w*h */
void orc_histogram(const uint8_t *img, unsigned w, unsigned h, unsigned hist[256]) {
  memset(hist, 0, 256 * sizeof(unsigned));
  unsigned n = w * h;
  for (unsigned i = 0; i < n; i++) hist[img[i]]++;
}

/* ref :205-223 -- float32 sequential scan; expression order is part of the contract */
uint8_t orc_otsu_from_hist(const unsigned hist[256], unsigned npix) {
  unsigned wb = 0, wf = 0, best = 0;
  float sum = 0, sumB = 0, varMax = -1.0f;
  for (unsigned i = 0; i < 256; i++) sum += (float)i * hist[i];
  for (unsigned t = 0; t < 256; t++) {
    wb += hist[t];
    if (wb == 0) continue;
    wf = npix - wb;
    if (wf == 0) break;
    sumB += (float)t * hist[t];
    float mB = sumB / wb;
    float mF = (sum - sumB) / wf;
    float between = (float)wb * (float)wf * (mB - mF) * (mB - mF);
    if (between > varMax) varMax = between, best = t;
  }
  return (uint8_t)best;
}

uint8_t orc_otsu_threshold(const uint8_t *img, unsigned w, unsigned h) {
  unsigned hist[256];
  orc_histogram(img, w, h, hist);
  return orc_otsu_from_hist(hist, w * h);
}

/* ref :225-228 */
void orc_threshold(uint8_t *img, unsigned w, unsigned h, uint8_t t) {
  unsigned n = w * h;
  for (unsigned i = 0; i < n; i++) img[i] = img[i] > t ? 255 : 0;
}

/* ref :230-247 -- clipped box mean minus c; `sum / count - c` is unsigned arithmetic
 * converted to int, then compared (as int) with the centre pixel */
void orc_adaptive_threshold(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                            unsigned radius, int c) {
  long r = (long)radius;
  for (long y = 0; y < (long)h; y++) {
    long ya = y - r < 0 ? 0 : y - r, yb = y + r > (long)h - 1 ? (long)h - 1 : y + r;
    for (long x = 0; x < (long)w; x++) {
      long xa = x - r < 0 ? 0 : x - r, xb = x + r > (long)w - 1 ? (long)w - 1 : x + r;
      unsigned sum = 0;
      for (long yy = ya; yy <= yb; yy++)
        for (long xx = xa; xx <= xb; xx++) sum += src[(size_t)yy * w + xx];
      unsigned count = (unsigned)((xb - xa + 1) * (yb - ya + 1));
      int thr = (int)(sum / count - (unsigned)c);
      dst[(size_t)y * w + x] = ((int)src[(size_t)y * w + x] > thr) ? 255 : 0;
    }
  }
}

/* ref :255-266 -- zero-padded correlation with an int8 kernel; `sum / norm` is
 * int/unsigned => evaluated in unsigned (negative sums with norm>1 saturate to 255) */
void orc_filter(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, const uint8_t *kernel,
                unsigned kw, unsigned kh, unsigned norm) {
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++) {
      int sum = 0;
      for (unsigned j = 0; j < kh; j++)
        for (unsigned i = 0; i < kw; i++)
          sum += px(src, w, h, x + i - kw / 2, y + j - kh / 2) * (int8_t)kernel[j * kw + i];
      sum = (int)((unsigned)sum / norm);
      dst[(size_t)y * w + x] = (uint8_t)(sum < 0 ? 0 : sum > 255 ? 255 : sum);
    }
}

/* ref :189-197 -- 2x2 mean, dst is (sw/2) x (sh/2) */
void orc_downsample(uint8_t *dst, const uint8_t *src, unsigned sw, unsigned sh) {
  unsigned dw = sw / 2, dh = sh / 2;
  for (unsigned y = 0; y < dh; y++)
    for (unsigned x = 0; x < dw; x++) {
      unsigned s = px(src, sw, sh, 2 * x, 2 * y) + px(src, sw, sh, 2 * x + 1, 2 * y) +
                   px(src, sw, sh, 2 * x, 2 * y + 1) + px(src, sw, sh, 2 * x + 1, 2 * y + 1);
      dst[(size_t)y * dw + x] = (uint8_t)(s / 4);
    }
}

/* ---------------------------------------------------------------- geometry + template matching */
/* ref :154-158 -- copy the roi (gs_get: 0 outside src; gs_set: dropped outside dst) */
void orc_crop(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src, unsigned sw, unsigned sh,
              unsigned rx, unsigned ry, unsigned rw, unsigned rh) {
  for (unsigned y = 0; y < rh; y++)
    for (unsigned x = 0; x < rw; x++)
      if (x < dw && y < dh) dst[(size_t)y * dw + x] = px(src, sw, sh, rx + x, ry + y);
}

/* ref :164-169 -- nearest neighbour, integer index arithmetic (unsigned, may wrap for huge sizes) */
void orc_resize_nn(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src, unsigned sw, unsigned sh) {
  for (unsigned y = 0; y < dh; y++)
    for (unsigned x = 0; x < dw; x++) {
      unsigned sx = x * sw / dw, sy = y * sh / dh;
      dst[(size_t)y * dw + x] = px(src, sw, sh, sx, sy);
    }
}

/* ref :171-187 -- bilinear, float32, pixel centres at +0.5; every operation is a float op in the
 * reference's order (unsigned -> float conversions included), the result truncates to uint8 */
void orc_resize(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src, unsigned sw, unsigned sh) {
  for (unsigned y = 0; y < dh; y++)
    for (unsigned x = 0; x < dw; x++) {
      float sx = ((float)x + 0.5f) * sw / dw - 0.5f;
      float sy = ((float)y + 0.5f) * sh / dh - 0.5f;
      float mx = sw - 1.0f, my = sh - 1.0f;
      sx = sx < mx ? sx : mx, sx = 0.0f > sx ? 0.0f : sx;
      sy = sy < my ? sy : my, sy = 0.0f > sy ? 0.0f : sy;
      unsigned xi = (unsigned)sx, yi = (unsigned)sy;
      unsigned x1 = xi + 1 < sw - 1 ? xi + 1 : sw - 1, y1 = yi + 1 < sh - 1 ? yi + 1 : sh - 1;
      float dx = sx - xi, dy = sy - yi;
      uint8_t c00 = px(src, sw, sh, xi, yi), c01 = px(src, sw, sh, x1, yi),
              c10 = px(src, sw, sh, xi, y1), c11 = px(src, sw, sh, x1, y1);
      uint8_t p = (c00 * (1 - dx) * (1 - dy)) + (c01 * dx * (1 - dy)) + (c10 * (1 - dx) * dy) +
                  (c11 * dx * dy);
      dst[(size_t)y * dw + x] = p;
    }
}

/* ref :705-724 -- sum of squared differences per offset, normalised: 255 - min(sum*255/max, 255) */
void orc_match_template(const uint8_t *img, unsigned iw, unsigned ih, const uint8_t *tmpl, unsigned tw,
                        unsigned th, uint8_t *result) {
  unsigned rw = iw - tw + 1, rh = ih - th + 1;
  unsigned long long max_diff = (unsigned long long)tw * th * 255ULL * 255ULL;
  for (unsigned ry = 0; ry < rh; ry++)
    for (unsigned rx = 0; rx < rw; rx++) {
      unsigned long long sum = 0;
      for (unsigned ty = 0; ty < th; ty++)
        for (unsigned tx = 0; tx < tw; tx++) {
          int d = (int)px(img, iw, ih, rx + tx, ry + ty) - (int)tmpl[(size_t)ty * tw + tx];
          sum += (unsigned long long)(d * d);
        }
      unsigned score = (unsigned)(sum * 255ULL / max_diff);
      result[(size_t)ry * rw + rx] = (uint8_t)(255 - (score < 255 ? score : 255));
    }
}

/* ref :726-739 -- first strict maximum in raster order; (0,0) when everything is 0 */
void orc_find_best_match(const uint8_t *result, unsigned w, unsigned h, unsigned *bx, unsigned *by) {
  uint8_t best = 0;
  *bx = 0, *by = 0;
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++)
      if (result[(size_t)y * w + x] > best) best = result[(size_t)y * w + x], *bx = x, *by = y;
}

/* ---------------------------------------------------------------- integral image */
/* ref :744-752 -- inclusive, same w x h shape, u32 modular */
void orc_integral(const uint8_t *src, unsigned w, unsigned h, unsigned *ii) {
  for (unsigned y = 0; y < h; y++) {
    unsigned run = 0;
    for (unsigned x = 0; x < w; x++) {
      run += src[(size_t)y * w + x];
      ii[(size_t)y * w + x] = run + (y ? ii[(size_t)(y - 1) * w + x] : 0u);
    }
  }
}

/* ref :754-763 -- D + A - B - C with x>0 / y>0 guards instead of a padded table */
unsigned orc_integral_sum(const unsigned *ii, unsigned iw, unsigned x, unsigned y, unsigned w,
                          unsigned h) {
  unsigned x2 = x + w - 1, y2 = y + h - 1;
  unsigned A = (x && y) ? ii[(size_t)(y - 1) * iw + (x - 1)] : 0;
  unsigned B = y ? ii[(size_t)(y - 1) * iw + x2] : 0;
  unsigned C = x ? ii[(size_t)y2 * iw + (x - 1)] : 0;
  unsigned D = ii[(size_t)y2 * iw + x2];
  return D + A - B - C;
}

/* ---------------------------------------------------------------- LBP cascade */
/* ref :769-783 -- 3x3 grid of fw x fh cells; bit order TL,TC,TR,R,BR,BC,BL,L (MSB->LSB),
 * bit = (cell >= centre) */
static int lbp_code(const unsigned *ii, unsigned iw, int x0, int y0, int fw, int fh) {
  unsigned cell[3][3];
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++)
      cell[j][i] = orc_integral_sum(ii, iw, (unsigned)(x0 + i * fw), (unsigned)(y0 + j * fh),
                                    (unsigned)fw, (unsigned)fh);
  unsigned c = cell[1][1];
  static const int order[8][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 2}, {2, 2}, {2, 1}, {2, 0}, {1, 0}};
  int code = 0;
  for (int k = 0; k < 8; k++) code = (code << 1) | (cell[order[k][0]][order[k][1]] >= c);
  return code;
}

/* ref :790-813 (+ :785-788 for the subset bit test) */
unsigned orc_lbp_window(const orc_cascade *c, const unsigned *ii, unsigned iw, unsigned ih, int x,
                        int y, float scale) {
  int win_w = (int)(c->window_w * scale), win_h = (int)(c->window_h * scale);
  if (x + win_w > (int)iw || y + win_h > (int)ih) return 0;
  for (int s = 0; s < c->nstages; s++) {
    int first = c->stage_weak_start[s], cnt = c->stage_nweaks[s];
    float sum = 0.0f; /* sequential float32 adds in weak order */
    for (int k = 0; k < cnt; k++) {
      int wi = first + k, fi = c->weak_feature_idx[wi];
      int fx = (int)(c->features[fi * 4 + 0] * scale);
      int fy = (int)(c->features[fi * 4 + 1] * scale);
      int fw = (int)(c->features[fi * 4 + 2] * scale);
      int fh = (int)(c->features[fi * 4 + 3] * scale);
      if (fw < 1) fw = 1;
      if (fh < 1) fh = 1;
      int code = lbp_code(ii, iw, x + fx, y + fy, fw, fh);
      const int32_t *sub = c->subsets + c->weak_subset_offset[wi];
      int word = code / 32, bit = code % 32;
      int hit = (word < (int)c->weak_num_subsets[wi]) && (((uint32_t)sub[word] >> bit) & 1u);
      sum += hit ? c->weak_left_val[wi] : c->weak_right_val[wi];
    }
    if (sum < c->stage_threshold[s]) return 0;
  }
  return 1;
}

/* ref :815-835 -- float32 scale progression; output order (scale, y, x); stop at max_rects */
unsigned orc_lbp_detect(const orc_cascade *c, const unsigned *ii, unsigned iw, unsigned ih,
                        orc_rect *rects, unsigned max_rects, float scale_factor, float min_scale,
                        float max_scale, int step) {
  unsigned n = 0;
  for (float scale = min_scale; scale <= max_scale && n < max_rects; scale *= scale_factor) {
    int win_w = (int)(c->window_w * scale), win_h = (int)(c->window_h * scale);
    if (win_w > (int)iw || win_h > (int)ih) break;
    for (int y = 0; y + win_h <= (int)ih && n < max_rects; y += step)
      for (int x = 0; x + win_w <= (int)iw && n < max_rects; x += step)
        if (orc_lbp_window(c, ii, iw, ih, x, y, scale)) {
          rects[n].x = (unsigned)x, rects[n].y = (unsigned)y;
          rects[n].w = (unsigned)win_w, rects[n].h = (unsigned)win_h;
          n++;
        }
  }
  return n;
}

/* ---------------------------------------------------------------- FAST-9 */
/* ref :482-534.  Pass 1: walk the 16-px ring for 16+9 steps with a signed run counter;
 * the comparisons are UNSIGNED (`p + threshold`, `p - threshold` with unsigned
 * threshold), so p < threshold makes the "darker" bound wrap.  Score = min|ring-p|
 * when a run of 9 exists, else 0; only the interior (3-px frame excluded) of the
 * scoremap is written.  Pass 2: strict-greater 3x3 NMS reading the scoremap
 * (including its never-written frame), raster-order emit, capped at nkps. */
unsigned orc_fast(const uint8_t *img, unsigned w, unsigned h, uint8_t *scoremap, orc_keypoint *kps,
                  unsigned nkps, unsigned threshold) {
  static const int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  static const int oy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  if (w < 7 || h < 7) return 0; /* reference loops are empty for 3<=dim<7 (and wrap below 3) */
  for (unsigned y = 3; y < h - 3; y++)
    for (unsigned x = 3; x < w - 3; x++) {
      unsigned p = img[(size_t)y * w + x];
      int run = 0, score = 0;
      for (int i = 0; i < 25; i++) {
        unsigned v = px(img, w, h, x + ox[i & 15], y + oy[i & 15]);
        if (v > p + threshold) run = run > 0 ? run + 1 : 1;
        else if (v < p - threshold) run = run < 0 ? run - 1 : -1;
        else run = 0;
        if (run >= 9 || run <= -9) {
          score = 255;
          for (int j = 0; j < 16; j++) {
            int d = (int)px(img, w, h, x + ox[j], y + oy[j]) - (int)p;
            if (d < 0) d = -d;
            if (d < score) score = d;
          }
          break;
        }
      }
      scoremap[(size_t)y * w + x] = (uint8_t)score;
    }
  unsigned n = 0;
  for (unsigned y = 3; y < h - 3; y++)
    for (unsigned x = 3; x < w - 3; x++) {
      int s = scoremap[(size_t)y * w + x];
      if (!s) continue;
      int is_max = 1;
      for (int dy = -1; dy <= 1 && is_max; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          if (!dx && !dy) continue;
          if (px(scoremap, w, h, x + dx, y + dy) > s) { is_max = 0; break; }
        }
      if (is_max && n < nkps) {
        memset(&kps[n], 0, sizeof kps[n]);
        kps[n].x = x, kps[n].y = y, kps[n].response = (unsigned)s;
        n++;
      }
    }
  return n;
}

/* ---------------------------------------------------------------- ORB */
/* The reference has two trig back-ends: libm (ref :100-101, what `make test` and nanomagick use) and,
 * under -DGS_NO_STDLIB, the two polynomials of ref :70-88 (what the wasm build uses,
 * examples/wasm/grayskull.c:32).  The polynomials are pure + - x / float32 and therefore reproducible
 * on the GPU; orc_set_nostdlib(1) switches this restatement to them. */
static int g_nostdlib = 0;
void orc_set_nostdlib(int on) { g_nostdlib = on != 0; }
float orc_atan2_poly(float y, float x) { /* ref :70-78 */
  if (x == 0.0f) return y > 0.0f ? 1.570796f : (y < 0.0f ? -1.570796f : 0.0f);
  float r, angle, abs_y = y >= 0.0f ? y : -y;
  if (x >= 0.0f) {
    r = (x - abs_y) / (x + abs_y);
    angle = 0.785398f - 0.785398f * r;
  } else {
    r = (x + abs_y) / (abs_y - x);
    angle = 3.0f * 0.785398f - 0.785398f * r;
  }
  return y < 0.0f ? -angle : angle;
}
float orc_sin_poly(float x) { /* ref :80-88 */
  while (x > 3.141592f) x -= 6.283185f;
  while (x < -3.141592f) x += 6.283185f;
  int sign = 1;
  if (x < 0) x = -x, sign = -1;
  if (x > 1.570796f) x = 3.141592f - x;
  float x2 = x * x, res = x * (1.0f - x2 * (0.16666667f - 0.0083333310f * x2));
  return sign * res;
}
static float trig_atan2(float y, float x) { return g_nostdlib ? orc_atan2_poly(y, x) : atan2f(y, x); }
static float trig_sin(float x) { return g_nostdlib ? orc_sin_poly(x) : sinf(x); }

/* ref :608-621 -- intensity-centroid moments over the disc dx^2+dy^2 <= r^2, float accum */
void orc_orientation_moments(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y,
                             unsigned r, float *m01, float *m10) {
  float a = 0, b = 0;
  for (int dy = -(int)r; dy <= (int)r; dy++)
    for (int dx = -(int)r; dx <= (int)r; dx++)
      if (dx * dx + dy * dy <= (int)(r * r)) {
        uint8_t I = px(img, w, h, x + dx, y + dy);
        a += dy * I;
        b += dx * I;
      }
  *m01 = a, *m10 = b;
}

float orc_orientation(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y,
                      unsigned r) {
  float m01, m10;
  orc_orientation_moments(img, w, h, x, y, r, &m01, &m10);
  return trig_atan2(m01, m10); /* ref :100 / :70, :620 */
}

/* ref :623-637 -- rotated BRIEF-256; cos = sin(angle + 1.57079f); float32, no FMA;
 * (int) truncation; out-of-image taps read 0; bit i set iff I1 > I2 */
void orc_brief(const uint8_t *img, unsigned w, unsigned h, orc_keypoint *kp) {
  int x = (int)kp->x, y = (int)kp->y;
  float ang = kp->angle, sn = trig_sin(ang), cs = trig_sin((float)(ang + 1.57079f));
  memset(kp->desc, 0, sizeof kp->desc);
  for (int i = 0; i < 256; i++) {
    const int8_t *q = &k_brief[i * 4];
    float ax = q[0] * cs - q[1] * sn, ay = q[0] * sn + q[1] * cs;
    float bx = q[2] * cs - q[3] * sn, by = q[2] * sn + q[3] * cs;
    uint8_t I1 = px(img, w, h, (unsigned)(x + (int)ax), (unsigned)(y + (int)ay));
    uint8_t I2 = px(img, w, h, (unsigned)(x + (int)bx), (unsigned)(y + (int)by));
    if (I1 > I2) kp->desc[i / 32] |= 1u << (i % 32);
  }
}

/* ref :639-649 -- stable, descending by response (insertion sort == the reference's bubble) */
static void sort_desc_stable(orc_keypoint *k, unsigned n) {
  for (unsigned i = 1; i < n; i++) {
    orc_keypoint t = k[i];
    unsigned j = i;
    while (j > 0 && k[j - 1].response < t.response) k[j] = k[j - 1], j--;
    k[j] = t;
  }
}

/* ref :651-669 -- FAST (cap min(4*nkps,5000)) -> sort -> 15-px border filter -> angle -> BRIEF */
unsigned orc_orb_extract(const uint8_t *img, unsigned w, unsigned h, orc_keypoint *kps,
                         unsigned nkps, unsigned threshold, uint8_t *scoremap) {
  unsigned cap = nkps * 4 < 5000 ? nkps * 4 : 5000;
  orc_keypoint *cand = (orc_keypoint *)malloc(sizeof(orc_keypoint) * 5000);
  unsigned nf = orc_fast(img, w, h, scoremap, cand, cap, threshold);
  sort_desc_stable(cand, nf);
  unsigned n = 0, r = 15;
  for (unsigned i = 0; i < nf && n < nkps; i++) {
    unsigned x = cand[i].x, y = cand[i].y;
    if (x >= r && y >= r && x < w - r && y < h - r) {
      kps[n] = cand[i];
      kps[n].angle = orc_orientation(img, w, h, x, y, r);
      orc_brief(img, w, h, &kps[n]);
      n++;
    }
  }
  free(cand);
  return n;
}

/* ref examples/nanomagick/nanomagick.c:245-290 (extract_pyramid_orb_nm) -- the ORB caller of the
 * reference: up to 4 levels, each half the size of the previous (ref grayskull.h:189-197), stop when
 * a level would be narrower or lower than 32; `buffer` holds levels 1.. back to back and then one
 * scoremap per level (the caller's bytes: the NMS reads their never-written 3-px frame);
 * nkps / n_levels keypoints per level, the last level takes the remainder; coordinates scaled back
 * by 2^level. */
unsigned orc_orb_extract_pyramid(const uint8_t *img, unsigned w, unsigned h, orc_keypoint *kps,
                                 unsigned nkps, unsigned threshold, uint8_t *buffer, unsigned n_levels) {
  if (n_levels > 4) n_levels = 4;
  const uint8_t *lev[4];
  unsigned lw[4], lh[4], total = 0, off = 0;
  lev[0] = img, lw[0] = w, lh[0] = h;
  for (unsigned l = 1; l < n_levels; l++) {
    unsigned nw = lw[l - 1] / 2, nh = lh[l - 1] / 2;
    if (nw < 32 || nh < 32) {
      n_levels = l;
      break;
    }
    uint8_t *d = buffer + off;
    off += nw * nh;
    orc_downsample(d, lev[l - 1], lw[l - 1], lh[l - 1]);
    lev[l] = d, lw[l] = nw, lh[l] = nh;
  }
  for (unsigned l = 0; l < n_levels; l++) {
    uint8_t *sm = buffer + off;
    off += lw[l] * lh[l];
    unsigned want = nkps / n_levels;
    if (l == n_levels - 1) want = nkps - total;
    if (want == 0) continue;
    unsigned got = orc_orb_extract(lev[l], lw[l], lh[l], &kps[total], want, threshold, sm);
    for (unsigned i = total; i < total + got; i++) kps[i].x <<= l, kps[i].y <<= l;
    total += got;
  }
  return total;
}

/* ref :671-699 -- brute force; best/second as float; accept iff best<=max && best<0.8*second */
unsigned orc_match_orb(const orc_keypoint *k1, unsigned n1, const orc_keypoint *k2, unsigned n2,
                       orc_match *out, unsigned max_matches, float max_distance) {
  unsigned n = 0;
  for (unsigned i = 0; i < n1 && n < max_matches; i++) {
    float best = max_distance + 1, second = max_distance + 1;
    unsigned arg = 0;
    for (unsigned j = 0; j < n2; j++) {
      unsigned bits = 0;
      for (int q = 0; q < 8; q++) bits += (unsigned)__builtin_popcount(k1[i].desc[q] ^ k2[j].desc[q]);
      float d = (float)bits;
      if (d < best) second = best, best = d, arg = j;
      else if (d < second) second = d;
    }
    if (best <= max_distance && best < 0.8f * second) {
      out[n].idx1 = i, out[n].idx2 = arg, out[n].distance = (unsigned)best;
      n++;
    }
  }
  return n;
}
