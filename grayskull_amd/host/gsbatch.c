/*
 * gsbatch -- multi-stage, multi-file PGM driver over libgrayskull_hip.so (C99 host code).
 *
 * SURVEY.md 8(f) rank 3.  The reference's CLI (examples/nanomagick/nanomagick.c:380-446) runs ONE
 * verb on ONE file per process; chains are shell pipes of PGM streams (reference Makefile:25-30),
 * i.e. every stage parses a header, allocates, runs and re-serialises the image.  This driver takes
 * the same verbs with the same arguments and the same validation, but
 *   - reads all inputs once, groups frames of equal size into batches,
 *   - uploads a batch once, keeps it in HBM between stages (two ping-pong planes),
 *   - runs every stage as one gsh_*_batch launch over the whole group (size-changing verbs: one
 *     stream-ordered gs_* call per frame on device pointers),
 *   - recognises `blur r : sobel [: threshold otsu]` (r = 1..3) and runs it through the fused
 *     one-pass kernels (gsh_blur_sobel_batch / gsh_edge_pipeline_batch),
 *   - downloads once and writes the results.
 * Per file the output bytes are identical to piping the reference's nanomagick through the same
 * verbs (tests/test_gsbatch.py).
 *
 * Verbs (nanomagick.c:52-141, same argument meaning / error text):
 *   resize <w> <h> | crop <x> <y> <w> <h> | blur <r> | threshold <t|otsu> | adaptive <r> <c> |
 *   sobel | morph <erode|dilate> <n>
 * and, as the LAST stage of a chain, the two feature verbs (nanomagick.c:217-243, :347-376):
 *   keypoints <n> <t>   gs_fast (cap 5000, threshold t) on the GPU; the n strongest are drawn as
 *                       crosses like nanomagick does and listed in <out>.keypoints.txt ("x y response")
 *   orb <template.pgm>  nanomagick's pyramid ORB matcher (nanomagick.c:245-345): 3-level gs_orb_extract of the
 *                       template and of the frame (2500 keypoints, threshold 20), gs_match_orb (300, 60.0),
 *                       the stitched picture with the 15 best matches drawn, and <out>.orb.txt
 *   faces <n>           gs_integral + gs_lbp_detect (cap 100, scale 1.2, 1..4, step n) on the GPU with
 *                       the cascade blob given by --cascade; boxes drawn like nanomagick, listed in
 *                       <out>.faces.txt ("x y w h").  (nanomagick's 640x480 limit is its static buffer
 *                       and is not imposed here.)
 * --gpus N shards the files of every size group over N GPUs (contiguous, balanced blocks of the
 * group's files: the frame_range rule of grayskull_amd/shard.py), one host thread per device
 * (gsh_set_device); frames never leave their GPU.  What crosses GPUs is SURVEY.md 8(e)'s control traffic, over RCCL
 * (gsh_comm_*, one communicator per worker from ncclCommInitAll; librccl is looked up at run time, one GPU takes local
 * copies and several GPUs without a usable librccl a host-rendezvous backend): the cascade blob is read by worker 0 and BROADCAST, every worker's per-file counts and output checksums
 * are ALL-GATHERED, the job's wall time is the ALL-REDUCE(max) of the workers'.  -v prints the checksum of checksums
 * over the files in command-line order -- for the same frames and chain the digest bench.py reports as
 * output_checksum_of_checksums.
 * PGM reading / writing follows the reference's gs_read_pgm / gs_write_pgm (grayskull.h:111-136):
 * binary P5, maxval 255, header "P5\n%u %u\n255\n".
 *
 * usage: gsbatch [-v] [--gpus N] [--cascade blob] -o <outdir> <verb> [args] [: <verb> [args]]... -- in1.pgm [in2.pgm ...]
 * exit:  0 all files written; 1 usage / stage error / at least one file failed (message on stderr,
 *        same wording as nanomagick where it has one).
 */
#define _POSIX_C_SOURCE 200112L /* clock_gettime, pthreads under -std=c99 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "grayskull_hip.h"

static size_t kSliceBytes = (size_t)64 << 20; /* per plane and slice (8 4K frames); GSBATCH_SLICE_BYTES overrides.
   Measured on 64 4K files (profiles/r01i_gsbatch_64x4k.log): page-locking the staging buffer costs ~85 ms
   per GiB, so small slices win: 0.44 s wall at 64 MiB vs 0.64 s at 1 GiB */

enum verb { V_RESIZE, V_CROP, V_BLUR, V_THRESHOLD, V_ADAPTIVE, V_SOBEL, V_MORPH, V_KEYPOINTS, V_FACES, V_ORB };
#define IS_TERMINAL(v) ((v) == V_KEYPOINTS || (v) == V_FACES || (v) == V_ORB)
enum { kFastCap = 5000, kFaceCap = 100, kOrbKps = 2500, kOrbMatches = 300 }; /* nanomagick.c:224, :349, :301-306 */

struct stage {
  enum verb v;
  int a[4];        /* numeric arguments in command-line order */
  int otsu;        /* threshold otsu */
  int dilate;      /* morph dilate */
  const char *raw; /* first argument as typed, for error messages */
};

static const struct {
  const char *name;
  enum verb v;
  int argc;
} verbs[] = {{"resize", V_RESIZE, 2},     {"crop", V_CROP, 4},   {"blur", V_BLUR, 1}, {"threshold", V_THRESHOLD, 1},
             {"adaptive", V_ADAPTIVE, 2}, {"sobel", V_SOBEL, 0}, {"morph", V_MORPH, 2},
             {"keypoints", V_KEYPOINTS, 2}, {"faces", V_FACES, 1}, {"orb", V_ORB, 1}, {NULL, V_SOBEL, 0}};

struct frame {
  const char *path;
  unsigned w, h;   /* input size */
  long data_off;   /* file offset of the first pixel */
  int group;       /* index of the (w,h) group, -1 = unreadable */
  int failed;
};

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void usage(const char *app) {
  fprintf(stderr,
          "Usage: %s [-v] [--gpus N] [--cascade blob] -o <outdir> <verb> [params] [: <verb> [params]]... -- in1.pgm [in2.pgm ...]\n"
          "Verbs: resize <w> <h> | crop <x> <y> <w> <h> | blur <r> | threshold <t|otsu> |\n"
          "       adaptive <r> <c> | sobel | morph <erode|dilate> <n>\n"
          "       last stage only: keypoints <n> <t> | faces <n> (needs --cascade) | orb <template.pgm>\n",
          app);
}

/* Header of a binary PGM as the reference reads it (grayskull.h:111-127: fscanf "P5\n%u %u\n%u\n",
 * maxval must be 255).  The reference's format string ends in "\n", which makes fscanf swallow EVERY
 * whitespace byte after maxval -- including leading pixels whose value happens to be 9..13 or 32;
 * such a file then comes up short and gs_read_pgm rejects it.  Here: when exactly w*h bytes follow
 * the single whitespace byte the PGM format prescribes, the raster starts there (identical to the
 * reference for every file it accepts, and the files it rejects for that reason load too);
 * otherwise the raster starts after all whitespace, like the reference. */
static int read_pgm_header(const char *path, struct frame *f) {
  FILE *fp = fopen(path, "rb");
  unsigned w, h, maxval;
  long after_one, end, after_all;
  int c;
  if (!fp) return -1;
  if (fscanf(fp, "P5 %u %u %u", &w, &h, &maxval) != 3 || maxval != 255 || w == 0 || h == 0) goto bad;
  c = fgetc(fp);
  if (c != ' ' && c != '\t' && c != '\n' && c != '\v' && c != '\f' && c != '\r') goto bad;
  after_one = ftell(fp);
  while ((c = fgetc(fp)) == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r') {}
  after_all = c == EOF ? ftell(fp) : ftell(fp) - 1;
  if (fseek(fp, 0, SEEK_END) != 0) goto bad;
  end = ftell(fp);
  fclose(fp);
  f->w = w, f->h = h;
  if ((unsigned long long)(end - after_one) == (unsigned long long)w * h) f->data_off = after_one;
  else if ((unsigned long long)(end - after_all) >= (unsigned long long)w * h) f->data_off = after_all;
  else return -1;
  return 0;
bad:
  fclose(fp);
  return -1;
}

static int read_pgm_pixels(const struct frame *f, uint8_t *dst) {
  FILE *fp = fopen(f->path, "rb");
  size_t n;
  if (!fp) return -1;
  if (fseek(fp, f->data_off, SEEK_SET) != 0) {
    fclose(fp);
    return -1;
  }
  n = fread(dst, 1, (size_t)f->w * f->h, fp);
  fclose(fp);
  return n == (size_t)f->w * f->h ? 0 : -1;
}

/* grayskull.h:129-136 */
static int write_pgm(const char *path, const uint8_t *data, unsigned w, unsigned h) {
  FILE *fp = fopen(path, "wb");
  size_t n;
  if (!fp) return -1;
  fprintf(fp, "P5\n%u %u\n255\n", w, h);
  n = fwrite(data, 1, (size_t)w * h, fp);
  fclose(fp);
  return n == (size_t)w * h ? 0 : -1;
}

static int parse_stages(int argc, char **argv, int *pos, struct stage *st, int max_stages) {
  int n = 0, i = *pos;
  while (i < argc && strcmp(argv[i], "--") != 0) {
    int k, j;
    if (strcmp(argv[i], ":") == 0) {
      i++;
      continue;
    }
    for (k = 0; verbs[k].name && strcmp(verbs[k].name, argv[i]) != 0; k++) {}
    if (!verbs[k].name) {
      fprintf(stderr, "Error: Unknown command '%s'\n", argv[i]);
      return -1;
    }
    if (n == max_stages) {
      fprintf(stderr, "Error: too many stages\n");
      return -1;
    }
    if (verbs[k].argc > 0 && i + verbs[k].argc >= argc) {
      fprintf(stderr, "Error: Wrong number of arguments for '%s'\n", argv[i]);
      return -1;
    }
    for (j = 0; j < verbs[k].argc; j++) {
      const char *s = argv[i + 1 + j];
      if (strcmp(s, ":") == 0 || strcmp(s, "--") == 0) {
        fprintf(stderr, "Error: Wrong number of arguments for '%s'\n", argv[i]);
        return -1;
      }
    }
    memset(&st[n], 0, sizeof st[n]);
    st[n].v = verbs[k].v;
    st[n].raw = verbs[k].argc ? argv[i + 1] : "";
    for (j = 0; j < verbs[k].argc; j++) st[n].a[j] = atoi(argv[i + 1 + j]);
    if (st[n].v == V_THRESHOLD) st[n].otsu = strcmp(argv[i + 1], "otsu") == 0;
    if (st[n].v == V_MORPH) {
      st[n].dilate = strcmp(argv[i + 1], "dilate") == 0;
      st[n].a[0] = (st[n].dilate || strcmp(argv[i + 1], "erode") == 0) ? 1 : 0; /* valid op */
    }
    i += 1 + verbs[k].argc;
    n++;
  }
  *pos = i;
  for (i = 0; i + 1 < n; i++)
    if (IS_TERMINAL(st[i].v)) {
      fprintf(stderr, "Error: '%s' writes detections, not an image to filter on: it must be the last stage\n",
              st[i].v == V_FACES ? "faces" : st[i].v == V_ORB ? "orb" : "keypoints");
      return -1;
    }
  return n;
}

/* nanomagick's per-verb argument checks (nanomagick.c:59-133); the image size is the group's */
static int check_stage(const struct stage *s, unsigned w, unsigned h) {
  switch (s->v) {
    case V_RESIZE:
      if (s->a[0] <= 0 || s->a[1] <= 0) return fprintf(stderr, "Error: Invalid width or height\n"), -1;
      break;
    case V_CROP:
      if (s->a[0] < 0 || s->a[1] < 0 || s->a[2] <= 0 || s->a[3] <= 0 || s->a[0] + s->a[2] > (int)w ||
          s->a[1] + s->a[3] > (int)h)
        return fprintf(stderr, "Error: Invalid crop rectangle\n"), -1;
      break;
    case V_BLUR:
      if (s->a[0] <= 0) return fprintf(stderr, "Error: Invalid radius: %s\n", s->raw), -1;
      break;
    case V_THRESHOLD:
      if (!s->otsu && s->a[0] <= 0) return fprintf(stderr, "Error: Invalid threshold: %s\n", s->raw), -1;
      break;
    case V_ADAPTIVE:
      if (s->a[0] <= 0 || s->a[1] < 0) return fprintf(stderr, "Error: Invalid radius or constant\n"), -1;
      break;
    case V_MORPH:
      if (!s->a[0] || s->a[1] <= 0)
        return fprintf(stderr, "Error: Invalid morphological operation or iterations\n"), -1;
      break;
    case V_KEYPOINTS: /* nanomagick.c:219-222 */
      if (s->a[0] <= 0 || s->a[1] < 0)
        return fprintf(stderr, "Error: Invalid number of keypoints or threshold\n"), -1;
      break;
    case V_FACES: /* nanomagick.c:351-355 */
      if (s->a[0] <= 0) return fprintf(stderr, "Error: minimum neighbors must be positive\n"), -1;
      break;
    case V_ORB:
    case V_SOBEL: break;
  }
  return 0;
}

static void stage_out_size(const struct stage *s, unsigned *w, unsigned *h) {
  if (s->v == V_RESIZE) *w = (unsigned)s->a[0], *h = (unsigned)s->a[1];
  if (s->v == V_CROP) *w = (unsigned)s->a[2], *h = (unsigned)s->a[3];
}

struct planes {
  uint8_t *cur, *other; /* device, each n * max frame bytes */
  unsigned *hist;       /* n * 256 u32 */
  uint8_t *thr_dev;     /* n */
  uint8_t *thr_host;    /* n */
};

static void swap_planes(struct planes *p) {
  uint8_t *t = p->cur;
  p->cur = p->other;
  p->other = t;
}

/* frames whose Otsu threshold is 0 fail exactly like `nanomagick threshold otsu` (nanomagick.c:89-93) */
static void otsu_failures(struct planes *p, unsigned n, int *failed) {
  unsigned f;
  gsh_download(p->thr_host, p->thr_dev, n);
  for (f = 0; f < n; f++)
    if (p->thr_host[f] == 0 && !failed[f]) {
      fprintf(stderr, "Error: Invalid threshold: otsu\n");
      failed[f] = 1;
    }
}

/* run the stage list over n device-resident frames of w x h; returns the output size */
static void run_stages(const struct stage *st, int ns, struct planes *p, unsigned n, unsigned *pw, unsigned *ph,
                       int *failed) {
  unsigned w = *pw, h = *ph, f;
  int i = 0;
  while (i < ns) {
    const struct stage *s = &st[i];
    const size_t fb = (size_t)w * h;
    /* blur r (1..3) : sobel [: threshold otsu] -> one pass over the frames */
    if (s->v == V_BLUR && s->a[0] <= 3 && i + 1 < ns && st[i + 1].v == V_SOBEL) {
      if (i + 2 < ns && st[i + 2].v == V_THRESHOLD && st[i + 2].otsu) {
        gsh_edge_pipeline_batch(p->other, NULL, p->cur, w, h, n, (unsigned)s->a[0], p->hist, p->thr_dev);
        otsu_failures(p, n, failed);
        i += 3;
      } else {
        gsh_blur_sobel_batch(p->other, p->cur, w, h, n, (unsigned)s->a[0]);
        i += 2;
      }
      swap_planes(p);
      continue;
    }
    switch (s->v) {
      case V_BLUR:
        gsh_blur_batch(p->other, p->cur, w, h, n, (unsigned)s->a[0]);
        swap_planes(p);
        break;
      case V_SOBEL: /* nanomagick's output image comes from calloc (nanomagick.c:138) */
        gsh_memset(p->other, 0, fb * n);
        if (w >= 3 && h >= 3) gsh_sobel_batch(p->other, p->cur, w, h, n);
        swap_planes(p);
        break;
      case V_THRESHOLD: /* copy + in-place threshold (nanomagick.c:94-96): the input is dead, so in place */
        if (s->otsu) {
          gsh_otsu_batch(p->cur, w, h, n, p->hist, p->thr_dev);
          otsu_failures(p, n, failed);
          gsh_threshold_batch_dev(p->cur, w, h, n, p->thr_dev);
        } else {
          gsh_threshold_batch(p->cur, w, h, n, (uint8_t)s->a[0]); /* (uint8_t) like gs_threshold's parameter */
        }
        break;
      case V_ADAPTIVE:
        gsh_adaptive_threshold_batch(p->other, p->cur, w, h, n, (unsigned)s->a[0], s->a[1]);
        swap_planes(p);
        break;
      case V_MORPH: {
        int it;
        for (it = 0; it < s->a[1]; it++) {
          if (s->dilate)
            gsh_dilate_batch(p->other, p->cur, w, h, n);
          else
            gsh_erode_batch(p->other, p->cur, w, h, n);
          swap_planes(p);
        }
        break;
      }
      case V_RESIZE:
      case V_CROP: {
        unsigned ow = w, oh = h;
        stage_out_size(s, &ow, &oh);
        for (f = 0; f < n; f++) {
          struct gs_image src = {w, h, p->cur + fb * f};
          struct gs_image dst = {ow, oh, p->other + (size_t)ow * oh * f};
          if (s->v == V_RESIZE) {
            gs_resize(dst, src);
          } else {
            struct gs_rect roi = {(unsigned)s->a[0], (unsigned)s->a[1], (unsigned)s->a[2], (unsigned)s->a[3]};
            gs_crop(dst, src, roi);
          }
        }
        w = ow;
        h = oh;
        swap_planes(p);
        break;
      }
      case V_KEYPOINTS:
      case V_FACES:
      case V_ORB: break; /* terminal verbs: handled by the worker after the image chain */
    }
    i++;
  }
  *pw = w;
  *ph = h;
}

static const char *base_name(const char *path) {
  const char *s = strrchr(path, '/');
  return s ? s + 1 : path;
}

/* ---- LBP cascade blob (layout: grayskull_amd/cascade.py; tests/golden/frontalface_cascade.bin) ---- */
struct cascade_file {
  struct gs_lbp_cascade c;
  uint8_t *raw;
  size_t raw_bytes;
};
static int cascade_tables_ok(const struct gs_lbp_cascade *c, uint32_t nsub) {
  /* every index array must stay inside the table it points into: gsh_cascade_create and the kernels trust them */
  unsigned i;
  if (c->window_w == 0 || c->window_h == 0) return 0;
  for (i = 0; i < c->nweaks; i++) {
    if (c->weak_feature_idx[i] >= c->nfeatures) return 0;
    if ((uint32_t)c->weak_subset_offset[i] + c->weak_num_subsets[i] > nsub) return 0;
  }
  for (i = 0; i < c->nstages; i++)
    if ((unsigned)c->stage_weak_start[i] + c->stage_nweaks[i] > c->nweaks) return 0;
  return 1;
}

/* the blob as bytes (read from a file by rank 0, received over RCCL by the others); takes ownership of raw */
static int parse_cascade(uint8_t *raw, size_t sz, struct cascade_file *cf) {
  uint16_t hdr[6];
  uint32_t nsub;
  size_t off = 20, cnt[10], esz[10] = {1, 2, 4, 4, 2, 2, 4, 2, 2, 4};
  const void *arr[10];
  int i;
  cf->raw = raw, cf->raw_bytes = sz;
  if (!raw || sz < 20) goto bad;
  if (memcmp(cf->raw, "LBPC", 4) != 0) goto bad;
  memcpy(hdr, cf->raw + 4, sizeof hdr); /* window_w, window_h, nfeatures, nweaks, nstages, pad */
  memcpy(&nsub, cf->raw + 16, 4);
  cnt[0] = (size_t)hdr[2] * 4, cnt[1] = cnt[2] = cnt[3] = cnt[4] = cnt[5] = hdr[3], cnt[6] = nsub;
  cnt[7] = cnt[8] = cnt[9] = hdr[4];
  for (i = 0; i < 10; i++) {
    arr[i] = cf->raw + off;
    if (cnt[i] > (sz - off) / esz[i]) goto bad; /* also keeps `off` from wrapping on a hostile nsub */
    off += (cnt[i] * esz[i] + 3) & ~(size_t)3;
    if (off > sz && i < 9) goto bad;
  }
  cf->c.window_w = hdr[0], cf->c.window_h = hdr[1], cf->c.nfeatures = hdr[2], cf->c.nweaks = hdr[3], cf->c.nstages = hdr[4];
  cf->c.features = (const int8_t *)arr[0], cf->c.weak_feature_idx = (const uint16_t *)arr[1];
  cf->c.weak_left_val = (const float *)arr[2], cf->c.weak_right_val = (const float *)arr[3];
  cf->c.weak_subset_offset = (const uint16_t *)arr[4], cf->c.weak_num_subsets = (const uint16_t *)arr[5];
  cf->c.subsets = (const int32_t *)arr[6], cf->c.stage_weak_start = (const uint16_t *)arr[7];
  cf->c.stage_nweaks = (const uint16_t *)arr[8], cf->c.stage_threshold = (const float *)arr[9];
  if (!cascade_tables_ok(&cf->c, nsub)) goto bad;
  return 0;
bad:
  free(cf->raw);
  cf->raw = NULL, cf->raw_bytes = 0;
  return -1;
}

static int load_cascade(const char *path, struct cascade_file *cf) {
  FILE *fp = fopen(path, "rb");
  long sz;
  uint8_t *raw;
  cf->raw = NULL, cf->raw_bytes = 0;
  if (!fp) return -1;
  if (fseek(fp, 0, SEEK_END) != 0 || (sz = ftell(fp)) < 20 || fseek(fp, 0, SEEK_SET) != 0) return fclose(fp), -1;
  raw = (uint8_t *)malloc((size_t)sz);
  if (!raw || fread(raw, 1, (size_t)sz, fp) != (size_t)sz) {
    fclose(fp);
    free(raw);
    return -1;
  }
  fclose(fp);
  return parse_cascade(raw, (size_t)sz, cf);
}

/* nanomagick.c:172-184, restated: Bresenham with clipping */
static void draw_line(uint8_t *img, unsigned w, unsigned h, unsigned x1, unsigned y1, unsigned x2, unsigned y2,
                      uint8_t color) {
  int dx = abs((int)x2 - (int)x1), dy = abs((int)y2 - (int)y1);
  int sx = x1 < x2 ? 1 : -1, sy = y1 < y2 ? 1 : -1, err = dx - dy;
  int x = (int)x1, y = (int)y1;
  for (;;) {
    int e2;
    if (x >= 0 && x < (int)w && y >= 0 && y < (int)h) img[(size_t)y * w + (unsigned)x] = color;
    if (x == (int)x2 && y == (int)y2) break;
    e2 = 2 * err;
    if (e2 > -dy) err -= dy, x += sx;
    if (e2 < dx) err += dx, y += sy;
  }
}
static void set_px(uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y) { /* gs_set: out of range = no-op */
  if (x < w && y < h) img[(size_t)y * w + x] = 255;
}
static int by_response_desc(const void *a, const void *b) { /* nanomagick.c:211-215, same comparator, same libc qsort */
  const struct gs_keypoint *k1 = (const struct gs_keypoint *)a, *k2 = (const struct gs_keypoint *)b;
  return (int)(k2->response - k1->response);
}

/* everything one worker (= one GPU) needs */
struct job {
  int device, ndev, verbose;
  const struct stage *st;
  int ns;
  struct frame *fr;
  int nf, ngroups;
  const char *outdir;
  const struct cascade_file *cascade; /* rank 0's copy (it read the file); the other ranks receive the bytes over RCCL */
  int rc;
  double t_alloc, t_read, t_up, t_run, t_down, t_write;
  /* multi-GPU control plane: one RCCL communicator per worker, created together by main() */
  gsh_comm *comm;
  uint64_t *file_sum, *file_cnt; /* nf entries each: checksum of the final plane / terminal-verb count of the files THIS worker owns */
  uint64_t *all_sum, *all_cnt;   /* nf entries each, filled on every worker by the all-gather: every file of the job */
  double t_wall, t_wall_max;     /* this worker's wall time; the job's (all-reduce max) */
};

/* contiguous, balanced share of `total` items for worker `rank` of `world` (grayskull_amd/shard.py frame_range) */
static void frame_range(unsigned rank, unsigned world, unsigned total, unsigned *lo, unsigned *hi) {
  const unsigned base = total / world, rem = total % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

/* A worker that left early would strand its peers inside the next collective (--gpus N: they would wait in RCCL for ever
 * and the process would hang in pthread_join instead of failing).  The only early exits were host allocation failures, so
 * they end the process the way the library ends on its own errors: message + abort(). */
static void gsb_oom(int line) {
  fprintf(stderr, "gsbatch: out of host memory (gsbatch.c:%d)\n", line);
  abort();
}

static void *worker(void *arg) {
  struct job *jb = (struct job *)arg;
  const struct stage *st = jb->st;
  const int ns = jb->ns, nf = jb->nf;
  struct frame *fr = jb->fr;
  const struct stage *term = (ns > 0 && IS_TERMINAL(st[ns - 1].v)) ? &st[ns - 1] : NULL;
  gsh_cascade *dc = NULL;
  int g, i;
  struct cascade_file mine; /* this worker's cascade, parsed from the bytes the broadcast delivered */
  const double t_start = now_ms();
  memset(&mine, 0, sizeof mine);
  gsh_set_device(jb->device);
  gsh_set_async(1); /* per-frame gs_resize / gs_crop on device pointers stay stream-ordered */
  if (term && term->v == V_FACES) {
    /* SURVEY 8(e) collective (1): rank 0 read the blob; its length travels as an all-reduce(max), its bytes as one
     * broadcast; every rank -- rank 0 included -- builds its device tables from the bytes it received */
    unsigned long long *len_dev = (unsigned long long *)gsh_malloc(8), len = jb->device == 0 ? jb->cascade->raw_bytes : 0;
    uint8_t *blob_dev, *blob;
    gsh_upload(len_dev, &len, 8);
    gsh_comm_all_reduce_u64(jb->comm, len_dev, 1, 1);
    gsh_download(&len, len_dev, 8);
    blob_dev = (uint8_t *)gsh_malloc((size_t)len);
    if (jb->device == 0) gsh_upload(blob_dev, jb->cascade->raw, (size_t)len);
    else gsh_memset(blob_dev, 0, (size_t)len);
    gsh_comm_broadcast(jb->comm, blob_dev, (size_t)len, 0);
    blob = (uint8_t *)malloc((size_t)len);
    if (!blob) gsb_oom(__LINE__);
    gsh_download(blob, blob_dev, (size_t)len);
    gsh_free(blob_dev), gsh_free(len_dev);
    if (parse_cascade(blob, (size_t)len, &mine) != 0) {
      fprintf(stderr, "Error: gpu %d received a cascade blob it cannot parse (%llu bytes)\n", jb->device, len);
      abort();
    }
    dc = gsh_cascade_create(&mine.c);
  }
  /* orb <template.pgm>: the template is read once and lives on the device */
  struct frame tmpl;
  uint8_t *tmpl_host = NULL, *tmpl_dev = NULL;
  struct gs_keypoint *tkps = NULL, *skps = NULL;
  struct gs_match *matches = NULL;
  memset(&tmpl, 0, sizeof tmpl);
  if (term && term->v == V_ORB) {
    tmpl.path = term->raw;
    if (read_pgm_header(tmpl.path, &tmpl) != 0 || !(tmpl_host = (uint8_t *)malloc((size_t)tmpl.w * tmpl.h)) ||
        read_pgm_pixels(&tmpl, tmpl_host) != 0) {
      if (jb->device == 0) printf("Error: Cannot load template image %s\n", tmpl.path); /* nanomagick.c:294, on stdout */
      tmpl.w = 0;
    } else {
      tmpl_dev = (uint8_t *)gsh_malloc((size_t)tmpl.w * tmpl.h);
      gsh_upload(tmpl_dev, tmpl_host, (size_t)tmpl.w * tmpl.h);
    }
    tkps = (struct gs_keypoint *)malloc(5000 * sizeof *tkps); /* nanomagick.c:301 */
    skps = (struct gs_keypoint *)malloc(5000 * sizeof *skps);
    matches = (struct gs_match *)malloc(kOrbMatches * sizeof *matches);
    if (!tkps || !skps || !matches) gsb_oom(__LINE__);
  }

  for (g = 0; g < jb->ngroups; g++) {
    unsigned w = 0, h = 0, ngroup = 0, n, lo, hi, f, ow, oh;
    size_t max_fb, fb;
    struct planes p;
    int *failed, bad = 0;
    uint8_t *stage;
    unsigned cap, b0;
    int *idx;
    double t0;
    /* terminal-verb buffers */
    uint8_t *orb_buf = NULL;
    size_t orb_buf_bytes = 0;
    uint8_t *score = NULL;
    uint64_t *sums_dev = NULL, *sums_host = NULL;
    struct gs_keypoint *kps_dev = NULL, *kps_host = NULL;
    unsigned *ii = NULL, *cnt_dev = NULL, *cnt_host = NULL;
    struct gs_rect *rects_dev = NULL, *rects_host = NULL;
    for (i = 0; i < nf; i++)
      if (fr[i].group == g) w = fr[i].w, h = fr[i].h, ngroup++;
    /* the share index rotates with the group so that many small groups (one odd-sized file each)
     * spread over the GPUs instead of all landing on GPU 0 */
    frame_range((unsigned)(jb->device + g) % (unsigned)jb->ndev, (unsigned)jb->ndev, ngroup, &lo, &hi);
    n = hi - lo;
    if (n == 0) continue;
    /* validate the chain for this size and find the largest plane it needs */
    ow = w, oh = h, max_fb = (size_t)w * h;
    for (i = 0; i < ns && !bad; i++) {
      if (check_stage(&st[i], ow, oh) != 0) bad = 1;
      stage_out_size(&st[i], &ow, &oh);
      if ((size_t)ow * oh > max_fb) max_fb = (size_t)ow * oh;
    }
    idx = (int *)malloc(ngroup * sizeof *idx);
    if (!idx) gsb_oom(__LINE__);
    for (i = 0, f = 0; i < nf; i++)
      if (fr[i].group == g) idx[f++] = i;
    if (bad) { /* nanomagick: the verb prints its message, then "did not produce output image" */
      for (f = lo; f < hi; f++) {
        fprintf(stderr, "Error: %s: chain did not produce output image\n", fr[idx[f]].path);
        fr[idx[f]].failed = 1;
      }
      jb->rc = 1;
      free(idx);
      continue;
    }
    fb = (size_t)w * h;
    /* a group is processed in slices of at most kSliceBytes per plane, so that the page-locked
     * staging buffer and the two device planes stay bounded whatever the number of files */
    cap = (unsigned)(kSliceBytes / max_fb);
    if (term && cap > 256) cap = 256; /* 240 KB of keypoint records / 4 bytes of integral per pixel per frame */
    cap = cap < 1 ? 1 : cap > n ? n : cap;
    t0 = now_ms();
    p.cur = (uint8_t *)gsh_malloc(max_fb * cap);
    p.other = (uint8_t *)gsh_malloc(max_fb * cap);
    p.hist = (unsigned *)gsh_malloc((size_t)cap * 256 * sizeof(unsigned));
    p.thr_dev = (uint8_t *)gsh_malloc(cap);
    p.thr_host = (uint8_t *)malloc(cap);
    failed = (int *)malloc(cap * sizeof *failed);
    stage = (uint8_t *)gsh_host_alloc(max_fb * cap); /* page-locked: one DMA each way per slice */
    sums_dev = (uint64_t *)gsh_malloc((size_t)cap * 8);
    sums_host = (uint64_t *)malloc((size_t)cap * 8);
    if (!sums_host) gsb_oom(__LINE__);
    if (term && term->v == V_ORB) {
      /* one scratch buffer for both pyramids, like nanomagick's (there: a static 1 MiB array, which a
       * frame beyond ~700x500 overruns; here: as large as the bigger pyramid needs) */
      const size_t a = tmpl.w ? gsh_orb_pyramid_buffer_bytes(tmpl.w, tmpl.h, 3) : 0, b = gsh_orb_pyramid_buffer_bytes(ow, oh, 3);
      orb_buf_bytes = a > b ? a : b;
      orb_buf = (uint8_t *)gsh_malloc(orb_buf_bytes);
    } else if (term) {
      const size_t ofb = (size_t)ow * oh;
      cnt_dev = (unsigned *)gsh_malloc((size_t)cap * sizeof(unsigned));
      cnt_host = (unsigned *)malloc((size_t)cap * sizeof(unsigned));
      if (term->v == V_KEYPOINTS) {
        score = (uint8_t *)gsh_malloc(ofb * cap);
        kps_dev = (struct gs_keypoint *)gsh_malloc((size_t)cap * kFastCap * sizeof *kps_dev);
        kps_host = (struct gs_keypoint *)malloc((size_t)cap * kFastCap * sizeof *kps_host);
      } else {
        ii = (unsigned *)gsh_malloc(ofb * cap * sizeof(unsigned));
        rects_dev = (struct gs_rect *)gsh_malloc((size_t)cap * kFaceCap * sizeof *rects_dev);
        rects_host = (struct gs_rect *)malloc((size_t)cap * kFaceCap * sizeof *rects_host);
      }
      if (!cnt_host || (term->v == V_KEYPOINTS ? !kps_host : !rects_host)) gsb_oom(__LINE__);
    }
    jb->t_alloc += now_ms() - t0;
    if (!p.thr_host || !failed) gsb_oom(__LINE__);

    for (b0 = 0; b0 < n; b0 += cap) {
      const unsigned nb = n - b0 < cap ? n - b0 : cap;
      memset(failed, 0, nb * sizeof *failed);
      t0 = now_ms();
      for (f = 0; f < nb; f++) {
        const struct frame *fi = &fr[idx[lo + b0 + f]];
        if (read_pgm_pixels(fi, stage + fb * f) != 0) {
          fprintf(stderr, "Error: Could not load %s\n", fi->path);
          memset(stage + fb * f, 0, fb);
          failed[f] = 2; /* unreadable: no second message later */
        }
      }
      jb->t_read += now_ms() - t0;

      t0 = now_ms();
      gsh_upload(p.cur, stage, fb * nb);
      jb->t_up += now_ms() - t0;

      t0 = now_ms();
      ow = w, oh = h;
      run_stages(st, ns, &p, nb, &ow, &oh, failed);
      if (term && term->v == V_KEYPOINTS) { /* nanomagick.c:229-230: gs_fast into a zeroed score map, cap 5000 */
        gsh_memset(score, 0, (size_t)ow * oh * nb);
        gsh_fast_batch(p.cur, score, ow, oh, nb, kps_dev, cnt_dev, kFastCap, (unsigned)term->a[1]);
      } else if (term && term->v == V_FACES) { /* nanomagick.c:362-364 */
        gsh_integral_batch(p.cur, ow, oh, nb, ii);
        gsh_lbp_detect_batch(dc, ii, ow, oh, nb, rects_dev, cnt_dev, kFaceCap, 1.2f, 1.0f, 4.0f, term->a[0]);
      }
      gsh_checksum_batch(p.cur, (size_t)ow * oh, nb, sums_dev); /* of the final plane, on the device (what bench.py checks) */
      gsh_sync();
      jb->t_run += now_ms() - t0;

      t0 = now_ms();
      gsh_download(stage, p.cur, (size_t)ow * oh * nb);
      gsh_download(sums_host, sums_dev, (size_t)nb * 8);
      if (term && term->v != V_ORB) {
        gsh_download(cnt_host, cnt_dev, (size_t)nb * sizeof(unsigned));
        if (term->v == V_KEYPOINTS) gsh_download(kps_host, kps_dev, (size_t)nb * kFastCap * sizeof *kps_host);
        else gsh_download(rects_host, rects_dev, (size_t)nb * kFaceCap * sizeof *rects_host);
      }
      jb->t_down += now_ms() - t0;

      t0 = now_ms();
      for (f = 0; f < nb; f++) {
        struct frame *fi = &fr[idx[lo + b0 + f]];
        uint8_t *img = stage + (size_t)ow * oh * f;
        char path[4096];
        if (failed[f]) {
          if (failed[f] == 1) fprintf(stderr, "Error: %s did not produce output image\n", fi->path);
          fi->failed = 1, jb->rc = 1;
          continue;
        }
        jb->file_sum[idx[lo + b0 + f]] = sums_host[f];
        jb->file_cnt[idx[lo + b0 + f]] = (term && term->v != V_ORB) ? cnt_host[f] : 0;
        if (term && term->v == V_ORB) { /* nanomagick.c:292-345, frame by frame like one process per file */
          unsigned nt, nsc, nm, k, x, y;
          FILE *rec;
          uint8_t *out;
          if (!tmpl.w) { /* the verb returned without an image (nanomagick.c:294-297) */
            fprintf(stderr, "Error: Command 'orb' did not produce output image\n");
            fi->failed = 1, jb->rc = 1;
            continue;
          }
          /* the scratch buffer starts zeroed in every nanomagick process and the scene's pyramid is built
           * over what the template's left behind (its never-written score-map frames are read by the NMS) */
          gsh_memset(orb_buf, 0, orb_buf_bytes);
          nt = gsh_orb_extract_pyramid(tmpl_dev, tmpl.w, tmpl.h, orb_buf, tkps, kOrbKps, 20, 3);
          nsc = gsh_orb_extract_pyramid(p.cur + (size_t)ow * oh * f, ow, oh, orb_buf, skps, kOrbKps, 20, 3);
          nm = gs_match_orb(tkps, nt, skps, nsc, matches, kOrbMatches, 60.0f);
          snprintf(path, sizeof path, "%s/%s.orb.txt", jb->outdir, base_name(fi->path));
          rec = fopen(path, "w");
          if (rec) fprintf(rec, "Template: %u keypoints, Scene: %u keypoints, Matches: %u\n", nt, nsc, nm);
          if (nm == 0) {
            if (rec) fclose(rec);
            fprintf(stderr, "Error: Command 'orb' did not produce output image\n"); /* nanomagick.c:431 */
            fi->failed = 1, jb->rc = 1;
            continue;
          }
          for (k = 0; k + 1 < nm; k++) { /* nanomagick.c:312-318: the same exchange sort, same order of ties */
            unsigned j;
            for (j = k + 1; j < nm; j++)
              if (matches[j].distance < matches[k].distance) {
                const struct gs_match t = matches[k];
                matches[k] = matches[j], matches[j] = t;
              }
          }
          for (k = 0; rec && k < nm; k++)
            fprintf(rec, "%u %u %u  %u %u  %u %u\n", matches[k].idx1, matches[k].idx2, matches[k].distance,
                    tkps[matches[k].idx1].pt.x, tkps[matches[k].idx1].pt.y, skps[matches[k].idx2].pt.x,
                    skps[matches[k].idx2].pt.y);
          if (rec) fclose(rec);
          { /* stitched picture: template left, frame right, the 15 best matches as lines (nanomagick.c:321-342) */
            const unsigned sw = tmpl.w + ow, sh = tmpl.h > oh ? tmpl.h : oh;
            out = (uint8_t *)calloc((size_t)sw * sh, 1);
            if (!out) gsb_oom(__LINE__);
            for (y = 0; y < tmpl.h; y++) memcpy(out + (size_t)y * sw, tmpl_host + (size_t)y * tmpl.w, tmpl.w);
            for (y = 0; y < oh; y++) memcpy(out + (size_t)y * sw + tmpl.w, img + (size_t)y * ow, ow);
            for (k = 0; k < (nm < 15 ? nm : 15); k++) {
              const unsigned x1 = tkps[matches[k].idx1].pt.x, y1 = tkps[matches[k].idx1].pt.y;
              x = skps[matches[k].idx2].pt.x + tmpl.w, y = skps[matches[k].idx2].pt.y;
              draw_line(out, sw, sh, x1, y1, x, y, 255);
            }
            snprintf(path, sizeof path, "%s/%s", jb->outdir, base_name(fi->path));
            if (write_pgm(path, out, sw, sh) != 0) {
              fprintf(stderr, "Error: Could not save %s\n", path);
              jb->rc = 1;
            }
            free(out);
          }
          continue;
        }
        if (term) { /* records first (the numbers before any drawing), then nanomagick's drawing */
          FILE *rec;
          unsigned k, cnt = cnt_host[f];
          snprintf(path, sizeof path, "%s/%s.%s.txt", jb->outdir, base_name(fi->path),
                   term->v == V_KEYPOINTS ? "keypoints" : "faces");
          rec = fopen(path, "w");
          if (!rec) {
            fprintf(stderr, "Error: Could not save %s\n", path);
            jb->rc = 1;
          }
          if (term->v == V_KEYPOINTS) {
            struct gs_keypoint *kp = kps_host + (size_t)f * kFastCap;
            const unsigned show = (unsigned)term->a[0] < cnt ? (unsigned)term->a[0] : cnt;
            qsort(kp, cnt, sizeof *kp, by_response_desc);
            for (k = 0; k < show; k++) {
              const unsigned x = kp[k].pt.x, y = kp[k].pt.y;
              int d;
              if (rec) fprintf(rec, "%u %u %u\n", x, y, kp[k].response);
              for (d = -2; d <= 2; d++) set_px(img, ow, oh, x, y + (unsigned)d), set_px(img, ow, oh, x + (unsigned)d, y);
            }
          } else {
            const struct gs_rect *r = rects_host + (size_t)f * kFaceCap;
            for (k = 0; k < cnt; k++) {
              if (rec) fprintf(rec, "%u %u %u %u\n", r[k].x, r[k].y, r[k].w, r[k].h);
              draw_line(img, ow, oh, r[k].x, r[k].y, r[k].x + r[k].w, r[k].y, 255);
              draw_line(img, ow, oh, r[k].x, r[k].y + r[k].h, r[k].x + r[k].w, r[k].y + r[k].h, 255);
              draw_line(img, ow, oh, r[k].x, r[k].y, r[k].x, r[k].y + r[k].h, 255);
              draw_line(img, ow, oh, r[k].x + r[k].w, r[k].y, r[k].x + r[k].w, r[k].y + r[k].h, 255);
            }
          }
          if (rec) fclose(rec);
        }
        snprintf(path, sizeof path, "%s/%s", jb->outdir, base_name(fi->path));
        if (write_pgm(path, img, ow, oh) != 0) {
          fprintf(stderr, "Error: Could not save %s\n", path);
          jb->rc = 1;
        }
      }
      jb->t_write += now_ms() - t0;
    }
    if (jb->verbose)
      fprintf(stderr, "gpu %d group %d: %u of %u frame(s) %ux%u -> %ux%u\n", jb->device, g, n, ngroup, w, h, ow, oh);
    gsh_free(p.cur);
    gsh_free(p.other);
    gsh_free(p.hist);
    gsh_free(p.thr_dev);
    gsh_host_free(stage);
    gsh_free(orb_buf);
    gsh_free(score);
    gsh_free(kps_dev);
    gsh_free(ii);
    gsh_free(rects_dev);
    gsh_free(cnt_dev);
    gsh_free(sums_dev);
    free(sums_host);
    free(kps_host);
    free(rects_host);
    free(cnt_host);
    free(p.thr_host);
    free(failed);
    free(idx);
  }
  if (dc) gsh_cascade_destroy(dc);
  free(mine.raw);
  gsh_free(tmpl_dev);
  free(tmpl_host);
  free(tkps);
  free(skps);
  free(matches);
  { /* SURVEY 8(e) collectives (2) and (4): every worker learns every file's count + checksum (all-gather of the nf-long
     * arrays, zero where a worker owns nothing: a file has one owner, so the column sums are the owners' values), and the
     * job's wall time is the all-reduce(max) of the workers' */
    const int world = gsh_comm_world(jb->comm);
    const size_t per = (size_t)2 * (size_t)nf * 8;
    uint64_t *send_host = (uint64_t *)malloc(per), *recv_host = (uint64_t *)malloc(per * (size_t)world);
    uint64_t *send_dev = (uint64_t *)gsh_malloc(per), *recv_dev = (uint64_t *)gsh_malloc(per * (size_t)world);
    double *t_dev = (double *)gsh_malloc(8);
    int r;
    if (!send_host || !recv_host) gsb_oom(__LINE__);
    memcpy(send_host, jb->file_sum, (size_t)nf * 8), memcpy(send_host + nf, jb->file_cnt, (size_t)nf * 8);
    gsh_upload(send_dev, send_host, per);
    gsh_comm_all_gather(jb->comm, send_dev, recv_dev, per);
    jb->t_wall = now_ms() - t_start;
    gsh_upload(t_dev, &jb->t_wall, 8);
    gsh_comm_all_reduce_f64(jb->comm, t_dev, 1, 1);
    gsh_download(recv_host, recv_dev, per * (size_t)world);
    gsh_download(&jb->t_wall_max, t_dev, 8);
    for (i = 0; i < nf; i++) {
      jb->all_sum[i] = jb->all_cnt[i] = 0;
      for (r = 0; r < world; r++)
        jb->all_sum[i] += recv_host[(size_t)r * 2 * (size_t)nf + (size_t)i], jb->all_cnt[i] += recv_host[(size_t)r * 2 * (size_t)nf + (size_t)nf + (size_t)i];
    }
    gsh_free(send_dev), gsh_free(recv_dev), gsh_free(t_dev);
    free(send_host), free(recv_host);
  }
  gsh_shutdown();
  return (void *)0;
}

int main(int argc, char **argv) {
  struct stage st[64];
  struct frame *fr;
  const char *outdir = NULL, *cascade_path = getenv("GSBATCH_CASCADE");
  struct cascade_file cf;
  struct job *jobs;
  gsh_comm **comms;
  pthread_t *th;
  int verbose = 0, pos = 1, ns, nf, i, ngroups = 0, rc = 0, ngpus = 1, d;
  double t_io0, t_read, t0;

  memset(&cf, 0, sizeof cf);
  while (pos < argc && argv[pos][0] == '-' && argv[pos][1] && strcmp(argv[pos], "--") != 0) {
    if (strcmp(argv[pos], "-v") == 0) {
      verbose = 1, pos++;
    } else if (strcmp(argv[pos], "-o") == 0 && pos + 1 < argc) {
      outdir = argv[pos + 1], pos += 2;
    } else if (strcmp(argv[pos], "--gpus") == 0 && pos + 1 < argc) {
      ngpus = atoi(argv[pos + 1]), pos += 2;
    } else if (strcmp(argv[pos], "--cascade") == 0 && pos + 1 < argc) {
      cascade_path = argv[pos + 1], pos += 2;
    } else if (strcmp(argv[pos], "--shard-table") == 0 && pos + 2 < argc) {
      /* diagnostic (no GPU needed): the file ranges `--gpus W` gives its workers for a group of T files, one
       * "rank lo hi" line each -- tests/test_shard.py holds them against grayskull_amd/shard.py frame_range */
      const unsigned W = (unsigned)atoi(argv[pos + 1]), T = (unsigned)atoi(argv[pos + 2]);
      unsigned r, lo, hi;
      if (W < 1) return 1;
      for (r = 0; r < W; r++) {
        frame_range(r, W, T, &lo, &hi);
        printf("%u %u %u\n", r, lo, hi);
      }
      return 0;
    } else {
      usage(argv[0]);
      return 1;
    }
  }
  ns = parse_stages(argc, argv, &pos, st, 64);
  if (ns < 0) return 1;
  if (ns == 0 || !outdir || ngpus < 1 || pos >= argc || strcmp(argv[pos], "--") != 0 || pos + 1 >= argc) {
    usage(argv[0]);
    return 1;
  }
  pos++;
  nf = argc - pos;
  if (st[ns - 1].v == V_FACES) {
    if (!cascade_path || load_cascade(cascade_path, &cf) != 0) {
      fprintf(stderr, "Error: faces needs a cascade blob (--cascade <file> or GSBATCH_CASCADE)%s%s\n",
              cascade_path ? ": cannot read " : "", cascade_path ? cascade_path : "");
      return 1;
    }
  }
  if (getenv("GSBATCH_SLICE_BYTES")) kSliceBytes = (size_t)strtoull(getenv("GSBATCH_SLICE_BYTES"), NULL, 10);
  fr = (struct frame *)calloc((size_t)nf, sizeof *fr);
  if (!fr) return 1;

  t_io0 = now_ms();
  for (i = 0; i < nf; i++) {
    int j;
    fr[i].path = argv[pos + i];
    fr[i].group = -1;
    if (read_pgm_header(fr[i].path, &fr[i]) != 0) {
      fprintf(stderr, "Error: Could not load %s\n", fr[i].path);
      fr[i].failed = 1;
      rc = 1;
      continue;
    }
    for (j = 0; j < i; j++)
      if (fr[j].group >= 0 && fr[j].w == fr[i].w && fr[j].h == fr[i].h) break;
    fr[i].group = j < i ? fr[j].group : ngroups++;
  }
  t_read = now_ms() - t_io0;

  d = gsh_device_count();
  if (d < 1) {
    fprintf(stderr, "Error: no HIP device\n");
    return 1;
  }
  if (ngpus > d) {
    fprintf(stderr, "Error: --gpus %d but only %d HIP device(s) visible\n", ngpus, d);
    return 1;
  }

  jobs = (struct job *)calloc((size_t)ngpus, sizeof *jobs);
  th = (pthread_t *)calloc((size_t)ngpus, sizeof *th);
  comms = (gsh_comm **)calloc((size_t)ngpus, sizeof *comms);
  if (!jobs || !th || !comms) return 1;
  /* one communicator per GPU, created together: RCCL (ncclCommInitAll; librccl is looked up at run time), local copies for
   * one GPU, a host-rendezvous backend when librccl cannot be used */
  if (gsh_comm_init_all(comms, ngpus, NULL) != 0) {
    fprintf(stderr, "Error: no communicator for --gpus %d\n", ngpus);
    return 1;
  }
  t0 = now_ms();
  for (d = 0; d < ngpus; d++) {
    struct job *jb = &jobs[d];
    jb->device = d, jb->ndev = ngpus, jb->verbose = verbose, jb->st = st, jb->ns = ns, jb->fr = fr, jb->nf = nf;
    jb->ngroups = ngroups, jb->outdir = outdir, jb->cascade = &cf, jb->comm = comms[d];
    jb->file_sum = (uint64_t *)calloc((size_t)nf * 4, 8);
    if (!jb->file_sum) return 1;
    jb->file_cnt = jb->file_sum + nf, jb->all_sum = jb->file_cnt + nf, jb->all_cnt = jb->all_sum + nf;
    if (ngpus == 1) {
      worker(jb); /* the calling thread is the one worker */
    } else if (pthread_create(&th[d], NULL, worker, jb) != 0) {
      fprintf(stderr, "Error: cannot start the worker for GPU %d\n", d);
      return 1;
    }
  }
  for (d = 0; d < ngpus; d++) {
    if (ngpus > 1) pthread_join(th[d], NULL);
    rc |= jobs[d].rc;
  }
  if (verbose) {
    /* what the collectives delivered, as worker 0 holds it: SURVEY 8(e)'s "checksum of checksums" over the files in
     * command-line order -- the digest bench.py prints as output_checksum_of_checksums for the same frames and chain */
    unsigned long long digest = 1469598103934665603ull, total = 0;
    for (i = 0; i < nf; i++) digest = (digest ^ jobs[0].all_sum[i]) * 1099511628211ull, total += jobs[0].all_cnt[i];
    fprintf(stderr, "collectives: %s | checksum of checksums %016llx over %d file(s), %llu result record(s) | job wall %.2f ms "
                    "(all-reduce max over %d worker(s))\n", gsh_comm_backend(comms[0]), digest, nf, total, jobs[0].t_wall_max, ngpus);
    fprintf(stderr, "files %d groups %d gpus %d | headers %.2f ms, workers %.2f ms wall\n", nf, ngroups, ngpus, t_read,
            now_ms() - t0);
    for (d = 0; d < ngpus; d++)
      fprintf(stderr, "gpu %d: alloc %.2f ms, read %.2f ms, upload %.2f ms, stages %.2f ms, download %.2f ms, write %.2f ms\n", d,
              jobs[d].t_alloc, jobs[d].t_read, jobs[d].t_up, jobs[d].t_run, jobs[d].t_down, jobs[d].t_write);
  }
  gsh_comm_destroy_all(comms, ngpus);
  for (d = 0; d < ngpus; d++) free(jobs[d].file_sum);
  free(comms);
  free(cf.raw);
  free(jobs);
  free(th);
  free(fr);
  return rc;
}
