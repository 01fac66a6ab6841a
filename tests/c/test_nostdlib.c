/*
 * tests/c/test_nostdlib.c -- a caller built the way examples/wasm/grayskull.c:31-35 builds the reference:
 * -DGS_NO_STDLIB before the header.  include/grayskull.h then binds gs_compute_orientation / gs_brief_descriptor /
 * gs_orb_extract to the polynomial-trig flavour (ref :70-88), and this program's output must equal the reference
 * header compiled with the same macro, bit for bit (tests/test_abi.py, tests/test_gpu_vs_reference.py).
 *
 *   test_nostdlib in.bin out.bin     in.bin: u32 w, h, nkps, threshold, then w*h bytes
 *                                    out.bin: u32 n, n keypoints (48 B), then for the first min(n, 8) keypoints the
 *                                    angle of a separate gs_compute_orientation call (f32) and the descriptor of a
 *                                    separate gs_brief_descriptor call (8 x u32); then the angles (f32) of four
 *                                    gs_compute_orientation calls closer than r to the border: (2, 3), (w - 1, h - 1),
 *                                    (0, 0), (w - 5, 7) -- the reference built this way has no assert (ref :69) and
 *                                    reads the pixels outside the image as 0 (gs_get, ref :41-43)
 * GS_NO_STDLIB drops the header's own <stdio.h>/<stdlib.h> (ref :68); this test harness includes them itself.
 */
#ifndef GS_NO_STDLIB
#error "build with -DGS_NO_STDLIB"
#endif
#include "grayskull.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
  unsigned hdr[4], n, i, m;
  struct gs_image img;
  struct gs_keypoint *kps;
  uint8_t *score;
  FILE *f;
  if (argc != 3) return 2;
  f = fopen(argv[1], "rb");
  if (!f || fread(hdr, 4, 4, f) != 4) return 3;
  img.w = hdr[0], img.h = hdr[1];
  img.data = (uint8_t *)malloc((size_t)img.w * img.h);
  score = (uint8_t *)calloc((size_t)img.w * img.h, 1);
  kps = (struct gs_keypoint *)calloc(hdr[2] ? hdr[2] : 1, sizeof *kps);
  if (!img.data || !score || !kps || fread(img.data, 1, (size_t)img.w * img.h, f) != (size_t)img.w * img.h) return 4;
  fclose(f);
  n = gs_orb_extract(img, kps, hdr[2], hdr[3], score); /* -> gs_orb_extract_nostdlib */
  f = fopen(argv[2], "wb");
  if (!f) return 5;
  fwrite(&n, 4, 1, f);
  fwrite(kps, sizeof *kps, n, f);
  m = n < 8 ? n : 8;
  for (i = 0; i < m; i++) {
    struct gs_keypoint k = kps[i];
    float a = gs_compute_orientation(img, k.pt.x, k.pt.y, 15); /* -> gs_compute_orientation_nostdlib */
    memset(k.descriptor, 0xff, sizeof k.descriptor);
    k.angle = a;
    gs_brief_descriptor(img, &k);                              /* -> gs_brief_descriptor_nostdlib */
    fwrite(&a, 4, 1, f);
    fwrite(k.descriptor, 4, 8, f);
  }
  {
    const unsigned bx[4] = {2, img.w - 1, 0, img.w - 5}, by[4] = {3, img.h - 1, 0, 7};
    for (i = 0; i < 4; i++) {
      float a = gs_compute_orientation(img, bx[i], by[i], 15);
      fwrite(&a, 4, 1, f);
    }
  }
  fclose(f);
  printf("n=%u\n", n);
  return 0;
}
